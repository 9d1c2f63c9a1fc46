// multi_gpu_driver.cpp -- the multi-GPU path of north_star from a C++ host, no Python: "one-map-replica-per-GPU across the
// node with RCCL broadcast of the voxel map over xGMI", queries sharded, no collective on the search path.
// This is the C++ twin of mpl_ros_amd/dist.py + bench.py --gpus N:
//   1. one RCCL communicator per visible GPU (ncclCommInitAll: one process, N devices -- a node launched as one process per
//      GPU uses ncclCommInitRank with an id passed over its launcher instead; everything below is the same per rank),
//   2. the map is generated on the host, uploaded to GPU 0 only, and ncclBroadcast puts a replica into every GPU's HBM,
//   3. every GPU's context ADOPTS its replica (mplx_map_set_device: not copied) and gets the same planner set-up,
//   4. the query stream is dealt longest-straight-line-first in a snake (dist.py partition "lpt"), one host thread per GPU
//      runs mplx_plan_batch on its share, the result rows are merged in stream order on the host,
//   5. check: GPU 0 plans the WHOLE stream alone; every row of the sharded run must equal it.
// usage: multi_gpu_driver [map edge = 128] [queries = 96]        prints one JSON line; exit code 0 iff every row matched
// build: hipcc -O2 -std=c++17 -I include tests/cpp/multi_gpu_driver.cpp mpl_ros_amd/csrc/libmplx.so -lrccl -lpthread
#include <hip/hip_runtime.h>
#include <mplx.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define HIPOK(x)                                                                          \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 3; } \
  } while (0)
#define NCCLOK(x)                                                                         \
  do {                                                                                    \
    ncclResult_t r_ = (x);                                                                \
    if (r_ != ncclSuccess) { printf("RCCL error %s at %s:%d\n", ncclGetErrorString(r_), __FILE__, __LINE__); return 4; } \
  } while (0)
#define MPLXOK(ctx, x)                                                                    \
  do {                                                                                    \
    int r_ = (x);                                                                         \
    if (r_ != MPLX_OK) { printf("mplx error %d: %s at %s:%d\n", r_, mplx_last_error(ctx), __FILE__, __LINE__); return 5; } \
  } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rng() {  // SplitMix64
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 128, nq = argc > 2 ? atoi(argv[2]) : 96;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { printf("no HIP device\n"); return 3; }
  const double res = 0.1, origin[3] = {0, 0, 0};
  const int32_t dim[3] = {n, n, n};
  const size_t cells = (size_t)n * n * n;
  // ---- the map: random boxes, 8 % of the volume (host, then GPU 0 only)
  std::vector<int8_t> grid(cells, 0);
  for (size_t filled = 0; filled < cells / 12;) {
    const int e[3] = {2 + (int)(rng() % 7), 2 + (int)(rng() % 7), 2 + (int)(rng() % 7)};
    const int o[3] = {(int)(rng() % (uint64_t)(n - e[0])), (int)(rng() % (uint64_t)(n - e[1])), (int)(rng() % (uint64_t)(n - e[2]))};
    for (int z = o[2]; z < o[2] + e[2]; z++)
      for (int y = o[1]; y < o[1] + e[1]; y++)
        for (int x = o[0]; x < o[0] + e[0]; x++) {
          int8_t &v = grid[(size_t)x + (size_t)n * y + (size_t)n * n * z];
          if (!v) { v = 100; filled++; }
        }
  }
  // ---- queries: pairs of free cell centres at least a quarter of the map apart
  std::vector<mplx_waypoint> S((size_t)nq), G((size_t)nq);
  auto free_point = [&](double *p) {
    for (;;) {
      const int c[3] = {(int)(rng() % (uint64_t)n), (int)(rng() % (uint64_t)n), (int)(rng() % (uint64_t)n)};
      if (grid[(size_t)c[0] + (size_t)n * c[1] + (size_t)n * n * c[2]]) continue;
      for (int k = 0; k < 3; k++) p[k] = (c[k] + 0.5) * res;
      return;
    }
  };
  for (int i = 0; i < nq; i++) {
    memset(&S[i], 0, sizeof(mplx_waypoint));
    memset(&G[i], 0, sizeof(mplx_waypoint));
    S[i].control = G[i].control = MPLX_ACC;
    for (;;) {
      free_point(S[i].pos);
      free_point(G[i].pos);
      double d = 0;
      for (int k = 0; k < 3; k++) d += (S[i].pos[k] - G[i].pos[k]) * (S[i].pos[k] - G[i].pos[k]);
      if (std::sqrt(d) >= 0.25 * n * res) break;
    }
  }
  // ---- 1. communicators and streams
  std::vector<int> devs((size_t)ndev);
  for (int d = 0; d < ndev; d++) devs[d] = d;
  std::vector<ncclComm_t> comms((size_t)ndev);
  NCCLOK(ncclCommInitAll(comms.data(), ndev, devs.data()));
  std::vector<hipStream_t> streams((size_t)ndev);
  std::vector<int8_t *> replica((size_t)ndev, nullptr);
  for (int d = 0; d < ndev; d++) {
    HIPOK(hipSetDevice(d));
    HIPOK(hipStreamCreate(&streams[d]));
    HIPOK(hipMalloc((void **)&replica[d], cells));
  }
  // ---- 2. upload to GPU 0, broadcast over xGMI
  HIPOK(hipSetDevice(0));
  HIPOK(hipMemcpy(replica[0], grid.data(), cells, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  HIPOK(hipEventCreate(&e0));
  HIPOK(hipEventCreate(&e1));
  HIPOK(hipEventRecord(e0, streams[0]));
  NCCLOK(ncclGroupStart());
  for (int d = 0; d < ndev; d++) NCCLOK(ncclBroadcast(replica[d], replica[d], cells, ncclInt8, 0, comms[d], streams[d]));
  NCCLOK(ncclGroupEnd());
  HIPOK(hipEventRecord(e1, streams[0]));
  for (int d = 0; d < ndev; d++) {
    HIPOK(hipSetDevice(d));
    HIPOK(hipStreamSynchronize(streams[d]));
  }
  float bcast_ms = 0;
  HIPOK(hipSetDevice(0));
  HIPOK(hipEventElapsedTime(&bcast_ms, e0, e1));
  // ---- 3. one context per GPU on its replica, same planner set-up (27-input acceleration lattice: map_planner_node.cpp:103-117)
  std::vector<double> U;
  for (int dx = -1; dx <= 1; dx++)
    for (int dy = -1; dy <= 1; dy++)
      for (int dz = -1; dz <= 1; dz++) { U.push_back(dx); U.push_back(dy); U.push_back(dz); }
  mplx_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.control = MPLX_ACC; cfg.n_u = 27; cfg.U = U.data();
  cfg.dt = 1.0; cfg.v_max = 2.0; cfg.a_max = 1.0; cfg.j_max = -1.0; cfg.w = 10.0; cfg.eps = 1.0;
  cfg.tol_pos = 0.5; cfg.tol_vel = -1.0; cfg.tol_acc = -1.0; cfg.t_max = INFINITY; cfg.max_expand = 200000; cfg.tol_yaw = -1.0;
  std::vector<mplx_ctx *> ctx((size_t)ndev, nullptr);
  for (int d = 0; d < ndev; d++) {
    MPLXOK(nullptr, mplx_ctx_create(d, &ctx[d]));
    MPLXOK(ctx[d], mplx_set_deadline(ctx[d], 120.0));
    MPLXOK(ctx[d], mplx_map_set_device(ctx[d], replica[d], dim, origin, res));
    MPLXOK(ctx[d], mplx_planner_config(ctx[d], &cfg));
    MPLXOK(ctx[d], mplx_set_capacity(ctx[d], std::min(nq, 256), (uint64_t)nq * 200000ull, (uint64_t)nq * 900000ull, (uint64_t)nq * 250000ull));
  }
  // ---- 4. deal the stream: longest straight-line distance first, snake over the GPUs (dist.py partition "lpt")
  std::vector<int> order((size_t)nq);
  for (int i = 0; i < nq; i++) order[i] = i;
  auto dist2 = [&](int i) { double d = 0; for (int k = 0; k < 3; k++) d += (S[i].pos[k] - G[i].pos[k]) * (S[i].pos[k] - G[i].pos[k]); return d; };
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return dist2(a) > dist2(b); });
  std::vector<std::vector<int>> part((size_t)ndev);
  for (int k = 0; k < nq; k++) {
    const int round = k / ndev, pos = k % ndev;
    part[(size_t)((round & 1) ? ndev - 1 - pos : pos)].push_back(order[k]);
  }
  std::vector<mplx_result> sharded((size_t)nq), whole((size_t)nq);
  std::vector<int> rc((size_t)ndev, 0);
  std::vector<std::thread> th;
  for (int d = 0; d < ndev; d++)
    th.emplace_back([&, d]() {
      const size_t m = part[d].size();
      if (!m) return;
      std::vector<mplx_waypoint> s(m), g(m);
      std::vector<mplx_result> r(m);
      for (size_t k = 0; k < m; k++) { s[k] = S[part[d][k]]; g[k] = G[part[d][k]]; }
      rc[d] = mplx_plan_batch(ctx[d], (int)m, s.data(), g.data(), r.data());
      for (size_t k = 0; k < m; k++) sharded[part[d][k]] = r[k];
    });
  for (auto &t : th) t.join();
  for (int d = 0; d < ndev; d++)
    if (rc[d] != MPLX_OK) { printf("mplx_plan_batch on GPU %d: %s\n", d, mplx_last_error(ctx[d])); return 5; }
  // ---- 5. the whole stream on GPU 0 alone; every row must match
  MPLXOK(ctx[0], mplx_plan_batch(ctx[0], nq, S.data(), G.data(), whole.data()));
  int bad = 0, ok_plans = 0;
  unsigned long long total_exp = 0;
  for (int i = 0; i < nq; i++) {
    const mplx_result &a = sharded[i], &b = whole[i];
    const bool same = a.status == b.status && a.traj_len == b.traj_len && a.n_expanded == b.n_expanded && a.n_nodes == b.n_nodes && a.expand_hash == b.expand_hash &&
                      (a.cost == b.cost || (std::isinf(a.cost) && std::isinf(b.cost)));
    bad += same ? 0 : 1;
    ok_plans += a.status == MPLX_PLAN_OK;
    total_exp += a.n_expanded;
  }
  printf("{\"gpus\": %d, \"map\": %d, \"queries\": %d, \"rccl_broadcast_ms\": %.3f, \"plans_ok\": %d, \"expansions\": %llu, \"rows_differing_from_one_gpu\": %d, \"per_gpu_queries\": [", ndev, n, nq,
         bcast_ms, ok_plans, total_exp, bad);
  for (int d = 0; d < ndev; d++) printf("%s%zu", d ? ", " : "", part[d].size());
  printf("]}\n");
  for (int d = 0; d < ndev; d++) {
    mplx_ctx_destroy(ctx[d]);
    HIPOK(hipSetDevice(d));
    HIPOK(hipFree(replica[d]));
    ncclCommDestroy(comms[d]);
  }
  return bad == 0 && ok_plans > 0 ? 0 : 1;
}
