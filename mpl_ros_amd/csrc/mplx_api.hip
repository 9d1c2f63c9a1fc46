// mplx_api.hip -- C-ABI (include/mplx.h) of the MI355X motion-primitive search back-end.
// Host side: context, map replica, planner configuration, device pools, launches, result copy-out.
// There is no CPU fallback: every compute entry point launches a gfx950 kernel or fails with
// MPLX_ERR_HIP.
#include "../../include/mplx.h"

#include <hip/hip_runtime.h>

#include <sched.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define MPLX_UTILITY_KERNELS
#include "mplx_kernels.h"
#include "mplx_poly_dev.h"

using namespace mplx;

static_assert(sizeof(SuccOut) == sizeof(mplx_succ), "SuccOut must mirror mplx_succ");
static_assert(sizeof(State) == 12 * sizeof(double), "State layout");

static std::string g_create_error;

struct mplx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = true;
  std::string err;
  // map
  int8_t *map = nullptr;
  bool own_map = false;
  uint32_t *bricks = nullptr;  // bit-packed occupancy in 8^3 bricks, rebuilt whenever the grid changes
  int8_t *aux = nullptr;       // potential field / search region (mplx_potential_* / mplx_search_region_set); null: none
  double pot_weight = 0.0;
  uint64_t aux_token = 0;      // whose auxiliary map this is (host wrappers sharing the context; 0: nobody's)
  int32_t dim[3] = {0, 0, 0};
  double origin[3] = {0, 0, 0};
  double res = 0;
  // config
  bool have_cfg = false;
  mplx_config cfg{};
  std::vector<double> U;
  double *dU = nullptr, *dUcost = nullptr;
  // yaw-carrying states (cfg.control had MPLX_YAW; cfg.control itself keeps the four derivative bits only)
  bool yaw = false;
  std::vector<double> Uyaw;  // per control input, zeros when the lattice has no yaw column
  double *dUyaw = nullptr, *d_traj_yaw = nullptr;
  bool last_yaw = false;
  std::vector<double> last_Uyaw;
  double bucket_width = 0;
  int speculation = -1;  // -1 auto, 0/1 off, else on
  // helper workgroups (look-ahead expansion on idle compute units): -1 auto (2 per leader), 0 off, 2
  int helpers = -1;
  int help_reserved = -1;   // workgroups that never lead in a batch larger than the machine (-1 auto: n_cus / 8 when nq >= 2 n_cus; 0 none)
  uint64_t help_rows = 0;   // rows of the heuristic cache (0 auto)
  int n_cus = 0;
  int pool_help_lanes = 0;  // unit width the heuristic-cache rows were sized for
  // capacities (shared by all queries of a batch)
  int32_t n_slots = 1;
  uint64_t cap_nodes = 1u << 20, cap_edges = 1u << 22, cap_log = 1u << 21;
  uint32_t cap_rec = 0;
  // pools
  bool pools_valid = false;
  int32_t pool_slots = 0, pool_control = 0;
  uint64_t pool_nodes = 0, pool_edges = 0, pool_log = 0;
  SearchParams pools{};  // only pool pointers / sizes are used
  std::vector<void *> pool_allocs;
  std::vector<uint32_t> last_node_table;
  // last batch (the getters answer from this snapshot, not from the current configuration)
  int last_nq = 0;
  bool last_single = false;
  int last_control = 0;
  double last_dt = 0;
  std::vector<double> last_U;
  uint64_t plan_epoch = 0;  // bumped by every mplx_plan / mplx_plan_batch
  bool help_cache_filled = false;  // (MPLX_DEBUG_KEEP_CACHE) a launch has left its look-ahead cache behind
  uint64_t map_epoch = 0, last_map_epoch = 0;  // bumped whenever the grid changes (build_bricks); value at the last plan
  std::vector<QueryOut> last_out;
  QueryOut *d_out = nullptr;
  QueryIn *d_in = nullptr;
  int32_t *d_traj_nodes = nullptr, *d_traj_actions = nullptr, *d_rec = nullptr, *d_next = nullptr, *d_order = nullptr;
  uint32_t *d_node_tables = nullptr, *d_edge_tables = nullptr;
  double *d_traj_states = nullptr;
  int batch_cap = 0;
  uint32_t batch_rec = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  float last_ms = 0;
  uint32_t help_epoch = 0;
  unsigned long long help_done_init = 0;
  HelpBox *dbg_boxes = nullptr;
  uint32_t help_stats[4] = {0, 0, 0, 0};
  uint32_t help_ctr_back[HELP_CTR_WORDS] = {};
  bool help_stats_pending = false;  // last batch: cache rows used, queries done, helpers that gave up on a stopped leader, helpers that found every leader served
  // streamed batches (mplx_plan_batch_submit / _wait): the batch launched on this context's stream and not yet collected.
  // The host-side sources of its asynchronous uploads live here until then.
  bool pending = false;
  int pend_nq = 0;
  bool pend_help = false;
  std::vector<QueryIn> pend_in;
  std::vector<int32_t> pend_order;
  // pool recycling (mplx_set_pool_recycling): the chunk bitmaps on the device, their initial (all free) image, and whether the last
  // batch ran with them (its queries' state spaces are gone then: the state-space getters refuse)
  bool recycle = false;
  uint32_t *d_chunk_bits = nullptr;
  std::vector<uint32_t> chunk_bits_init;
  uint32_t chunk_word0[3] = {0, 0, 0}, chunk_words[3] = {0, 0, 0};
  bool last_recycled = false;
  bool pool_recycle = false;  // what the pools were allocated for (the table is sized for more states than the pool holds at once)
  unsigned long long *table_base = nullptr;  // the shared state table as allocated
  uint32_t tbl_epoch = TBL_EPOCHS;           // epoch of the table's slots of the last launch (table_prepare); TBL_EPOCHS: clear before use
  uint32_t launch_count = 0;
  int help_limit = -1;  // workgroups of a launch that may turn into helpers once the query queue is empty (-1: all of them)
  // launch guard (mplx_device.h GuardBlock): host-coherent block the search kernels poll / leave their watch records in
  GuardBlock *guard = nullptr;
  // a search launch older than this is aborted (mplx_set_deadline, MPLX_DEADLINE_S).  OPT-IN: the default (<= 0) waits for ever, like the
  // reference's plan(), whose only bound is max_num (map_planner_node.cpp:186-196); tests and bench.py set one.
  double deadline_s = 0.0;
  std::chrono::steady_clock::time_point guard_t0;  // when the guarded launch was armed (guard_arm): the deadline counts from the LAUNCH, not from the wait
  bool wedged = false;        // a launch did not even answer the abort word: the context's stream is lost
  bool debug_hang = false;    // (tests) the next search launch spins until the host aborts it
  uint64_t cfg_epoch = 0;     // bumped by every change of the planner set-up / pool policy (an mplx_stream's lanes follow it)
  // expansion filter of the next launches (internal: getSubStateSpace by import, mplx_lpa.inl); null: none
  const unsigned long long *filter_table = nullptr;
  unsigned long long filter_mask = 0;
  const char *filter_pool = nullptr;
  uint32_t filter_flag = 0;
};
#define MPLX_REFUSE_PENDING(c)                                                                                                       \
  do {                                                                                                                               \
    if ((c)->pending) return fail((c), MPLX_ERR_ARG, "a submitted batch is still outstanding on this context (mplx_plan_batch_wait first)"); \
  } while (0)

static int fail(mplx_ctx *c, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_create_error = buf;
  return code;
}
#define HIPCHK(c, call)                                                                         \
  do {                                                                                          \
    hipError_t e__ = (call);                                                                    \
    if (e__ != hipSuccess) return fail((c), MPLX_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e__)); \
  } while (0)

// temporary device buffers of one entry point: freed on every exit path
struct DevBufs {
  std::vector<void *> v;
  ~DevBufs() {
    for (void *p : v) (void)hipFree(p);
  }
  template <typename T>
  hipError_t alloc(T **p, size_t bytes) {
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, bytes);
    if (e == hipSuccess) {
      v.push_back(q);
      *p = (T *)q;
    }
    return e;
  }
};

extern "C" const char *mplx_version(void) { return "mplx 0.1 (gfx950)"; }
extern "C" const char *mplx_last_error(const mplx_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

extern "C" int mplx_ctx_create(int device, mplx_ctx **out) {
  if (!out) return fail(nullptr, MPLX_ERR_ARG, "out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return fail(nullptr, MPLX_ERR_HIP, "no HIP device available (%s)", hipGetErrorString(e));
  if (device < 0 || device >= n) return fail(nullptr, MPLX_ERR_ARG, "device %d out of range (%d devices)", device, n);
  mplx_ctx *c = new mplx_ctx();
  c->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess ||
      hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
    delete c;
    return fail(nullptr, MPLX_ERR_HIP, "stream/event creation failed");
  }
  {
    hipDeviceProp_t prop;
    c->n_cus = hipGetDeviceProperties(&prop, device) == hipSuccess ? prop.multiProcessorCount : 256;
  }
  // launch guard: one block of host-coherent (fine-grained, mapped) memory -- the kernels reach it over the fabric, the host
  // reads and writes it as plain memory while a launch is running
  if (hipHostMalloc((void **)&c->guard, sizeof(GuardBlock), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
    c->guard = nullptr;
    mplx_ctx_destroy(c);
    return fail(nullptr, MPLX_ERR_HIP, "hipHostMalloc of the launch-guard block failed");
  }
  memset(c->guard, 0, sizeof(GuardBlock));
  if (const char *e = getenv("MPLX_DEADLINE_S")) c->deadline_s = atof(e);
  *out = c;
  return MPLX_OK;
}

static void free_pools(mplx_ctx *c) {
  for (void *p : c->pool_allocs) (void)hipFree(p);
  c->pool_allocs.clear();
  c->pools_valid = false;
}
static void free_batch(mplx_ctx *c) {
  (void)hipFree(c->d_out); (void)hipFree(c->d_in); (void)hipFree(c->d_traj_nodes); (void)hipFree(c->d_traj_actions);
  (void)hipFree(c->d_traj_states); (void)hipFree(c->d_traj_yaw); (void)hipFree(c->d_rec); (void)hipFree(c->d_next); (void)hipFree(c->d_order);
  (void)hipFree(c->d_node_tables); (void)hipFree(c->d_edge_tables);
  c->d_out = nullptr; c->d_in = nullptr; c->d_traj_nodes = c->d_traj_actions = c->d_rec = c->d_next = c->d_order = nullptr;
  c->d_node_tables = nullptr; c->d_edge_tables = nullptr;
  c->d_traj_states = c->d_traj_yaw = nullptr;
  c->batch_cap = 0;
}

extern "C" void mplx_ctx_destroy(mplx_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->wedged) {  // a launch that never ended still owns the stream and the pools: freeing them would block for ever -- leak them
    delete c;
    return;
  }
  (void)hipStreamSynchronize(c->stream);
  if (c->guard) (void)hipHostFree(c->guard);
  free_pools(c);
  free_batch(c);
  if (c->own_map) (void)hipFree(c->map);
  (void)hipFree(c->bricks);
  (void)hipFree(c->aux);
  (void)hipFree(c->dU);
  (void)hipFree(c->dUcost);
  (void)hipFree(c->dUyaw);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" int mplx_set_stream(mplx_ctx *c, void *s) {
  if (!c) return MPLX_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  if (s) {
    c->stream = (hipStream_t)s;
    c->own_stream = false;
  } else {
    HIPCHK(c, hipStreamCreate(&c->stream));
    c->own_stream = true;
  }
  return MPLX_OK;
}

// ------------------------------------------------------------------ map
static int build_bricks(mplx_ctx *c);
static int set_map_meta(mplx_ctx *c, const int32_t dim[3], const double origin[3], double res) {
  if (!dim || !origin || dim[0] <= 0 || dim[1] <= 0 || dim[2] <= 0 || !(res > 0)) return fail(c, MPLX_ERR_ARG, "bad map geometry");
  if (c->aux && (dim[0] != c->dim[0] || dim[1] != c->dim[1] || dim[2] != c->dim[2])) {  // another grid: the auxiliary map goes
    (void)hipFree(c->aux);
    c->aux = nullptr;
    c->aux_token = 0;  // nobody's any more: the planner that built it must build it again (its token would say "still mine")
  }
  for (int i = 0; i < 3; i++) {
    c->dim[i] = dim[i];
    c->origin[i] = origin[i];
  }
  c->res = res;
  return MPLX_OK;
}
extern "C" int mplx_map_set(mplx_ctx *c, const int8_t *data, const int32_t dim[3], const double origin[3], double res) {
  if (!c || !data) return fail(c, MPLX_ERR_ARG, "null argument");
  HIPCHK(c, hipSetDevice(c->device));
  int r = set_map_meta(c, dim, origin, res);
  if (r) return r;
  size_t n = (size_t)dim[0] * dim[1] * dim[2];
  if (c->own_map) (void)hipFree(c->map);
  c->map = nullptr;
  c->own_map = true;
  HIPCHK(c, hipMalloc((void **)&c->map, n));
  HIPCHK(c, hipMemcpyAsync(c->map, data, n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return build_bricks(c);
}
extern "C" int mplx_map_set_device(mplx_ctx *c, const void *dptr, const int32_t dim[3], const double origin[3], double res) {
  if (!c || !dptr) return fail(c, MPLX_ERR_ARG, "null argument");
  HIPCHK(c, hipSetDevice(c->device));
  int r = set_map_meta(c, dim, origin, res);
  if (r) return r;
  if (c->own_map) (void)hipFree(c->map);
  c->map = (int8_t *)dptr;
  c->own_map = false;
  return build_bricks(c);  // the caller guarantees the buffer is complete (e.g. broadcast + synchronize done)
}
extern "C" int mplx_map_free_unknown(mplx_ctx *c) {
  if (!c || !c->map) return fail(c, MPLX_ERR_ARG, "no map");
  HIPCHK(c, hipSetDevice(c->device));
  size_t n = (size_t)c->dim[0] * c->dim[1] * c->dim[2];
  hipLaunchKernelGGL(free_unknown_kernel, dim3(2048), dim3(256), 0, c->stream, c->map, n);
  c->map_epoch++;  // (unknown -> free changes is_free(start), not the occupancy bitmap)
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MPLX_OK;
}
extern "C" int mplx_map_get(mplx_ctx *c, int8_t *out) {
  if (!c || !c->map || !out) return fail(c, MPLX_ERR_ARG, "no map");
  HIPCHK(c, hipSetDevice(c->device));
  size_t n = (size_t)c->dim[0] * c->dim[1] * c->dim[2];
  HIPCHK(c, hipMemcpyAsync(out, c->map, n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MPLX_OK;
}
extern "C" int mplx_map_info(const mplx_ctx *c, int32_t dim[3], double origin[3], double *res) {
  if (!c || !c->map) return MPLX_ERR_ARG;
  for (int i = 0; i < 3; i++) {
    if (dim) dim[i] = c->dim[i];
    if (origin) origin[i] = c->origin[i];
  }
  if (res) *res = c->res;
  return MPLX_OK;
}
// (re)build the bricked occupancy bitmap from the byte grid
static int build_bricks(mplx_ctx *c) {
  c->map_epoch++;  // every change of the occupancy ends here (set, adopt, dilate, grid -> map)
  const int nb0 = (c->dim[0] + 7) / 8, nb1 = (c->dim[1] + 7) / 8, nb2 = (c->dim[2] + 7) / 8;
  (void)hipFree(c->bricks);
  c->bricks = nullptr;
  const size_t nwords = (size_t)nb0 * nb1 * nb2 * 16;
  if (nwords > ((size_t)1 << 30)) return fail(c, MPLX_ERR_ARG, "map too large: more than 2^26 bricks (2^35 voxels)");  // kernels index the bitmap with 32 bits
  HIPCHK(c, hipMalloc((void **)&c->bricks, nwords * sizeof(uint32_t)));
  hipLaunchKernelGGL(brick_pack_kernel, dim3(4096), dim3(256), 0, c->stream, c->map, c->dim[0], c->dim[1], c->dim[2], nb0, nb1, nb2, c->bricks);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MPLX_OK;
}

static MapDev map_dev(const mplx_ctx *c) {
  MapDev m;
  m.data = c->map;
  m.bricks = c->bricks;
  m.aux = c->aux;
  m.nb[0] = (c->dim[0] + 7) / 8; m.nb[1] = (c->dim[1] + 7) / 8; m.nb[2] = (c->dim[2] + 7) / 8;
  for (int i = 0; i < 3; i++) {
    m.dim[i] = c->dim[i];
    m.origin[i] = c->origin[i];
  }
  m.res = c->res;
  return m;
}
extern "C" int mplx_map_dilate(mplx_ctx *c, int n_off, const int32_t *off) {
  if (!c || !c->map || n_off < 0 || (n_off > 0 && !off)) return fail(c, MPLX_ERR_ARG, "bad argument");
  if (n_off == 0) return MPLX_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = (size_t)c->dim[0] * c->dim[1] * c->dim[2];
  int8_t *out = nullptr;
  int32_t *doff = nullptr;
  HIPCHK(c, hipMalloc((void **)&out, n));
  if (hipMalloc((void **)&doff, sizeof(int32_t) * 3 * (size_t)n_off) != hipSuccess) {
    (void)hipFree(out);
    return fail(c, MPLX_ERR_HIP, "hipMalloc failed");
  }
  hipError_t e = hipMemcpyAsync(doff, off, sizeof(int32_t) * 3 * (size_t)n_off, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(dilate_kernel, dim3(8192), dim3(256), 0, c->stream, c->map, out, c->dim[0], c->dim[1], c->dim[2], n_off, doff);
    e = hipGetLastError();
  }
  if (e == hipSuccess) {
    if (c->own_map) {  // swap buffers
      e = hipStreamSynchronize(c->stream);
      if (e == hipSuccess) {
        (void)hipFree(c->map);
        c->map = out;
        out = nullptr;
      }
    } else {           // the caller's buffer: results go back into it
      e = hipMemcpyAsync(c->map, out, n, hipMemcpyDeviceToDevice, c->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
  }
  (void)hipFree(out);
  (void)hipFree(doff);
  if (e != hipSuccess) return fail(c, MPLX_ERR_HIP, "%s", hipGetErrorString(e));
  return build_bricks(c);
}
extern "C" int mplx_map_cells(mplx_ctx *c, int n, const int32_t *cells, int8_t *state) {
  if (!c || !c->map || n < 0 || (n > 0 && (!cells || !state))) return fail(c, MPLX_ERR_ARG, "bad argument");
  if (n == 0) return MPLX_OK;
  HIPCHK(c, hipSetDevice(c->device));
  int32_t *dc = nullptr;
  int8_t *ds = nullptr;
  HIPCHK(c, hipMalloc((void **)&dc, sizeof(int32_t) * 3 * (size_t)n));
  if (hipMalloc((void **)&ds, (size_t)n) != hipSuccess) {
    (void)hipFree(dc);
    return fail(c, MPLX_ERR_HIP, "hipMalloc failed");
  }
  hipError_t e = hipMemcpyAsync(dc, cells, sizeof(int32_t) * 3 * (size_t)n, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(map_cells_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, map_dev(c), n, dc, ds);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(state, ds, (size_t)n, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(dc);
  (void)hipFree(ds);
  if (e != hipSuccess) return fail(c, MPLX_ERR_HIP, "%s", hipGetErrorString(e));
  return MPLX_OK;
}
// host-side geometry: reads no voxel (same f64 expressions as the device's float_to_cell)
extern "C" int mplx_map_raytrace(const mplx_ctx *c, const double p1[3], const double p2[3], int32_t *cells, int cap, int *n_out) {
  if (!c || !c->map || !p1 || !p2 || !n_out || cap < 0 || (cap > 0 && !cells)) return MPLX_ERR_ARG;
  double diff[3], m = 0.0;
  for (int i = 0; i < 3; i++) {
    diff[i] = p2[i] - p1[i];
    const double q = fabs(diff[i] / c->res);
    if (q > m) m = q;
  }
  const double k = 0.8;
  const int max_diff = (int)(m / k);
  const double sstep = 1.0 / max_diff;
  int32_t prev[3] = {-1, -1, -1};
  int cnt = 0;
  for (int n = 1; n < max_diff; n++) {
    int32_t pn[3];
    bool outside = false;
    for (int i = 0; i < 3; i++) {
      const double pt = p1[i] + diff[i] * sstep * n;
      pn[i] = float_to_cell(pt, c->origin[i], c->res);
      if (pn[i] < 0 || pn[i] >= c->dim[i]) outside = true;
    }
    if (outside) break;
    if (pn[0] != prev[0] || pn[1] != prev[1] || pn[2] != prev[2]) {
      if (cnt < cap) {
        cells[3 * cnt] = pn[0]; cells[3 * cnt + 1] = pn[1]; cells[3 * cnt + 2] = pn[2];
      }
      cnt++;
    }
    prev[0] = pn[0]; prev[1] = pn[1]; prev[2] = pn[2];
  }
  *n_out = cnt;
  return MPLX_OK;
}
extern "C" int mplx_map_cloud(mplx_ctx *c, int which, double *pts, uint64_t cap, uint64_t *n_out) {
  if (!c || !c->map || which < 0 || which > 2 || !n_out || (cap > 0 && !pts)) return fail(c, MPLX_ERR_ARG, "bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  const int ncol = c->dim[0] * c->dim[1];
  uint32_t *counts = nullptr;
  unsigned long long *offs = nullptr, *dtotal = nullptr;
  double *dpts = nullptr;
  hipError_t e = hipMalloc((void **)&counts, sizeof(uint32_t) * (size_t)ncol);
  if (e == hipSuccess) e = hipMalloc((void **)&offs, sizeof(unsigned long long) * (size_t)ncol);
  if (e == hipSuccess) e = hipMalloc((void **)&dtotal, sizeof(unsigned long long));
  unsigned long long total = 0;
  if (e == hipSuccess) {
    hipLaunchKernelGGL(cloud_count_kernel, dim3((ncol + 255) / 256 < 4096 ? (ncol + 255) / 256 : 4096), dim3(256), 0, c->stream, map_dev(c), which, counts);
    hipLaunchKernelGGL(cloud_scan_kernel, dim3(1), dim3(1024), 0, c->stream, counts, offs, ncol, dtotal);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(&total, dtotal, sizeof(total), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  const uint64_t nw = total < cap ? total : cap;
  if (e == hipSuccess && nw > 0) {
    e = hipMalloc((void **)&dpts, sizeof(double) * 3 * nw);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(cloud_write_kernel, dim3((ncol + 255) / 256 < 4096 ? (ncol + 255) / 256 : 4096), dim3(256), 0, c->stream, map_dev(c), which, offs, (unsigned long long)nw, dpts);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(pts, dpts, sizeof(double) * 3 * nw, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  }
  (void)hipFree(counts); (void)hipFree(offs); (void)hipFree(dtotal); (void)hipFree(dpts);
  if (e != hipSuccess) return fail(c, MPLX_ERR_HIP, "%s", hipGetErrorString(e));
  *n_out = total;
  return MPLX_OK;
}
extern "C" int mplx_map_query(mplx_ctx *c, int n, const double *pts, int32_t *cells, int8_t *state) {
  if (!c || !c->map || n < 0 || !pts || !cells || !state) return fail(c, MPLX_ERR_ARG, "bad argument");
  if (n == 0) return MPLX_OK;
  HIPCHK(c, hipSetDevice(c->device));
  double *dp = nullptr;
  int32_t *dc = nullptr;
  int8_t *ds = nullptr;
  DevBufs bufs;
  HIPCHK(c, bufs.alloc(&dp, sizeof(double) * 3 * n));
  HIPCHK(c, bufs.alloc(&dc, sizeof(int32_t) * 3 * n));
  HIPCHK(c, bufs.alloc(&ds, n));
  HIPCHK(c, hipMemcpyAsync(dp, pts, sizeof(double) * 3 * n, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(map_query_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, map_dev(c), n, dp, dc, ds);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(cells, dc, sizeof(int32_t) * 3 * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(state, ds, n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MPLX_OK;
}

// ------------------------------------------------------------------ potential field / search region
// MapPlanner::setPotentialRadius / setPotentialWeight / setGradientWeight / setPotentialMapRange / updatePotentialMap /
// setSearchRadius / setSearchRegion / getPotentialCloud / getSearchRegion (distance_map_planner_node.cpp:185-193,199,
// 218-224,231).  The implementation upstream is un-vendored; the semantics P1-P3 restated for the tests' CPU checker
// are the ones built here.
static int aux_ensure(mplx_ctx *c) {
  if (c->aux) return MPLX_OK;
  const size_t n = (size_t)c->dim[0] * c->dim[1] * c->dim[2];
  HIPCHK(c, hipMalloc((void **)&c->aux, n));
  HIPCHK(c, hipMemsetAsync(c->aux, 0, n, c->stream));
  return MPLX_OK;
}
extern "C" int mplx_potential_weights(mplx_ctx *c, double potential_weight, double gradient_weight) {
  if (!c) return MPLX_ERR_ARG;
  if (gradient_weight != 0.0) return fail(c, MPLX_ERR_ARG, "setGradientWeight(%g): only 0 (the value the reference passes) is supported", gradient_weight);
  c->pot_weight = potential_weight;
  return MPLX_OK;
}
extern "C" int mplx_aux_token(mplx_ctx *c, uint64_t set_value, int32_t do_set, uint64_t *current) {
  if (!c) return MPLX_ERR_ARG;
  if (do_set) c->aux_token = set_value;
  if (current) *current = c->aux_token;
  return MPLX_OK;
}
extern "C" int mplx_potential_clear(mplx_ctx *c) {
  if (!c) return MPLX_ERR_ARG;
  c->aux_token = 0;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  (void)hipFree(c->aux);
  c->aux = nullptr;
  c->map_epoch++;
  return MPLX_OK;
}
extern "C" int mplx_potential_update(mplx_ctx *c, const double radius[3], const double pos[3], const double range[3], int32_t pw) {
  if (!c || !c->map || !radius || !pos) return fail(c, MPLX_ERR_ARG, "bad argument / no map");
  HIPCHK(c, hipSetDevice(c->device));
  int r = aux_ensure(c);
  if (r) return r;
  if (pw < 1) pw = 1;
  // the mask H(n) = trunc(100 (1 - d)^pow), d = sqrt(sum (n_i res / radius_i)^2) <= 1
  int rn[3];
  for (int i = 0; i < 3; i++) rn[i] = radius[i] > 0 ? (int)ceil(radius[i] / c->res) : 0;
  std::vector<int32_t> off;
  std::vector<int8_t> val;
  for (int nx = -rn[0]; nx <= rn[0]; nx++)
    for (int ny = -rn[1]; ny <= rn[1]; ny++)
      for (int nz = -rn[2]; nz <= rn[2]; nz++) {
        const int nn[3] = {nx, ny, nz};
        double s = 0.0;
        for (int i = 0; i < 3; i++)
          if (radius[i] > 0) {
            const double q = (double)nn[i] * c->res / radius[i];
            s += q * q;
          }
        const double d = sqrt(s);
        if (d > 1.0) continue;
        const double v = 1.0 - d;
        double rr = v;
        for (int k = 1; k < pw; k++) rr = rr * v;
        const int h = (int)(100.0 * rr);
        if (h <= 0) continue;
        off.push_back(nx); off.push_back(ny); off.push_back(nz);
        val.push_back((int8_t)(h > 100 ? 100 : h));
      }
  PotArgs a{};
  for (int i = 0; i < 3; i++) { a.pos[i] = pos[i]; a.range[i] = range ? range[i] : 0.0; }
  a.n_mask = (int32_t)val.size();
  int32_t *doff = nullptr;
  int8_t *dval = nullptr;
  DevBufs bufs;
  HIPCHK(c, bufs.alloc(&doff, sizeof(int32_t) * std::max<size_t>(off.size(), 3)));
  HIPCHK(c, bufs.alloc(&dval, std::max<size_t>(val.size(), 1)));
  if (!val.empty()) {
    HIPCHK(c, hipMemcpyAsync(doff, off.data(), sizeof(int32_t) * off.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dval, val.data(), val.size(), hipMemcpyHostToDevice, c->stream));
  }
  hipLaunchKernelGGL(pot_update_kernel, dim3(4096), dim3(256), 0, c->stream, map_dev(c), c->aux, a, doff, dval);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->map_epoch++;
  return MPLX_OK;
}
extern "C" int mplx_search_region_set(mplx_ctx *c, int n_pts, const double *pts, const double radius[3], int dense) {
  if (!c || !c->map || n_pts < 0 || (n_pts > 0 && (!pts || !radius))) return fail(c, MPLX_ERR_ARG, "bad argument / no map");
  HIPCHK(c, hipSetDevice(c->device));
  int r = aux_ensure(c);
  if (r) return r;
  const size_t n = (size_t)c->dim[0] * c->dim[1] * c->dim[2];
  c->map_epoch++;
  if (n_pts == 0) {
    hipLaunchKernelGGL(region_apply_kernel, dim3(4096), dim3(256), 0, c->stream, n, (const int8_t *)nullptr, c->aux);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MPLX_OK;
  }
  // seed cells: the path's own cells, joined up with rayTrace when the path is sparse (geometry only: host code, like mplx_map_raytrace)
  std::vector<int32_t> seeds;
  for (int k = 0; k < n_pts; k++) {
    int32_t pn[3];
    bool outside = false;
    for (int i = 0; i < 3; i++) {
      pn[i] = float_to_cell(pts[3 * k + i], c->origin[i], c->res);
      if (pn[i] < 0 || pn[i] >= c->dim[i]) outside = true;
    }
    if (!outside) seeds.insert(seeds.end(), pn, pn + 3);
    if (!dense && k + 1 < n_pts) {
      int cnt = 0;
      mplx_map_raytrace(c, pts + 3 * k, pts + 3 * (k + 1), nullptr, 0, &cnt);
      if (cnt > 65535) cnt = 65535;
      const size_t at = seeds.size();
      seeds.resize(at + 3 * (size_t)cnt);
      int got = 0;
      if (cnt) mplx_map_raytrace(c, pts + 3 * k, pts + 3 * (k + 1), seeds.data() + at, cnt, &got);
    }
  }
  int rn[3];
  for (int i = 0; i < 3; i++) rn[i] = radius[i] > 0 ? (int)ceil(radius[i] / c->res) : 0;
  int8_t *in = nullptr;
  int32_t *dseeds = nullptr;
  DevBufs bufs;
  HIPCHK(c, bufs.alloc(&in, n));
  HIPCHK(c, bufs.alloc(&dseeds, sizeof(int32_t) * std::max<size_t>(seeds.size(), 3)));
  HIPCHK(c, hipMemsetAsync(in, 0, n, c->stream));
  const int n_seeds = (int)(seeds.size() / 3);
  if (n_seeds) {
    HIPCHK(c, hipMemcpyAsync(dseeds, seeds.data(), sizeof(int32_t) * seeds.size(), hipMemcpyHostToDevice, c->stream));
    const long long total = (long long)(2 * rn[0] + 1) * (2 * rn[1] + 1) * (2 * rn[2] + 1) * n_seeds;
    hipLaunchKernelGGL(region_mark_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, c->dim[0], c->dim[1], c->dim[2], n_seeds, dseeds, rn[0], rn[1], rn[2], in);
    HIPCHK(c, hipGetLastError());
  }
  hipLaunchKernelGGL(region_apply_kernel, dim3(4096), dim3(256), 0, c->stream, n, (const int8_t *)in, c->aux);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MPLX_OK;
}
extern "C" int mplx_aux_get(mplx_ctx *c, int8_t *out) {
  if (!c || !c->map || !out) return fail(c, MPLX_ERR_ARG, "no map");
  const size_t n = (size_t)c->dim[0] * c->dim[1] * c->dim[2];
  if (!c->aux) { memset(out, 0, n); return MPLX_OK; }
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(out, c->aux, n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MPLX_OK;
}
// which 0: voxels with a potential strictly between 0 and 100 (getPotentialCloud); 1: voxels of the search region
// (getSearchRegion).  pts: voxel centres, x outermost like getCloud; vals: the potential of each (may be NULL).
extern "C" int mplx_aux_cloud(mplx_ctx *c, int which, double *pts, int8_t *vals, uint64_t cap, uint64_t *n_out) {
  if (!c || !c->map || which < 0 || which > 1 || !n_out || (cap > 0 && !pts)) return fail(c, MPLX_ERR_ARG, "bad argument");
  *n_out = 0;
  if (!c->aux) return MPLX_OK;
  HIPCHK(c, hipSetDevice(c->device));
  MapDev m = map_dev(c);
  m.data = c->aux;  // the cloud kernels read `data`
  const int w = which == 0 ? 3 : 4;
  const int ncol = c->dim[0] * c->dim[1];
  uint32_t *counts = nullptr;
  unsigned long long *offs = nullptr, *dtotal = nullptr;
  double *dpts = nullptr;
  int8_t *dvals = nullptr;
  DevBufs bufs;
  HIPCHK(c, bufs.alloc(&counts, sizeof(uint32_t) * (size_t)ncol));
  HIPCHK(c, bufs.alloc(&offs, sizeof(unsigned long long) * (size_t)ncol));
  HIPCHK(c, bufs.alloc(&dtotal, sizeof(unsigned long long)));
  const int grid = (ncol + 255) / 256 < 4096 ? (ncol + 255) / 256 : 4096;
  hipLaunchKernelGGL(cloud_count_kernel, dim3(grid), dim3(256), 0, c->stream, m, w, counts);
  hipLaunchKernelGGL(cloud_scan_kernel, dim3(1), dim3(1024), 0, c->stream, counts, offs, ncol, dtotal);
  HIPCHK(c, hipGetLastError());
  unsigned long long total = 0;
  HIPCHK(c, hipMemcpyAsync(&total, dtotal, sizeof(total), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *n_out = total;
  const uint64_t nw = total < cap ? total : cap;
  if (nw > 0) {
    HIPCHK(c, bufs.alloc(&dpts, sizeof(double) * 3 * nw));
    HIPCHK(c, bufs.alloc(&dvals, nw));
    hipLaunchKernelGGL(cloud_write_kernel, dim3(grid), dim3(256), 0, c->stream, m, w, offs, (unsigned long long)nw, dpts, dvals);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(pts, dpts, sizeof(double) * 3 * nw, hipMemcpyDeviceToHost, c->stream));
    if (vals) HIPCHK(c, hipMemcpyAsync(vals, dvals, nw, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return MPLX_OK;
}

// ------------------------------------------------------------------ configuration
static bool control_ok(int c) { return c == CTRL_VEL || c == CTRL_ACC || c == CTRL_JRK || c == CTRL_SNP; }

extern "C" int mplx_planner_config(mplx_ctx *c, const mplx_config *cfg) {
  if (!c || !cfg || !cfg->U) return fail(c, MPLX_ERR_ARG, "null argument");
  MPLX_REFUSE_PENDING(c);  // (the batch in flight reads dU / dUcost)
  if (!control_ok(cfg->control & ~MPLX_YAW)) return fail(c, MPLX_ERR_ARG, "unsupported control %d", cfg->control);
  if (cfg->n_u <= 0 || cfg->n_u > 256) return fail(c, MPLX_ERR_ARG, "n_u must be in [1,256], got %d", cfg->n_u);
  if (!(cfg->dt > 0)) return fail(c, MPLX_ERR_ARG, "dt must be > 0");
  HIPCHK(c, hipSetDevice(c->device));
  c->cfg = *cfg;
  c->yaw = (cfg->control & MPLX_YAW) != 0;
  c->cfg.control = cfg->control & ~MPLX_YAW;
  cfg = &c->cfg;
  c->U.assign(cfg->U, cfg->U + 3 * (size_t)cfg->n_u);
  c->cfg.U = c->U.data();
  c->Uyaw.assign((size_t)cfg->n_u, 0.0);
  if (c->yaw && cfg->U_yaw) c->Uyaw.assign(cfg->U_yaw, cfg->U_yaw + cfg->n_u);
  c->cfg.U_yaw = c->Uyaw.data();
  // per-control edge cost J(control) + w dt: depends on the control input only
  std::vector<double> ucost(cfg->n_u);
  for (int i = 0; i < cfg->n_u; i++) {
    double cc[3][6];
    for (int ax = 0; ax < 3; ax++) prim_build_axis(cfg->control, 0.0, 0.0, 0.0, 0.0, c->U[3 * i + ax], cc[ax]);
    ucost[i] = prim_J(cfg->control, cc, cfg->dt) + cfg->w * cfg->dt;
  }
  (void)hipFree(c->dU);
  (void)hipFree(c->dUcost);
  (void)hipFree(c->dUyaw);
  c->dU = c->dUcost = c->dUyaw = nullptr;
  if (c->yaw) {
    HIPCHK(c, hipMalloc((void **)&c->dUyaw, sizeof(double) * cfg->n_u));
    HIPCHK(c, hipMemcpyAsync(c->dUyaw, c->Uyaw.data(), sizeof(double) * cfg->n_u, hipMemcpyHostToDevice, c->stream));
  }
  HIPCHK(c, hipMalloc((void **)&c->dU, sizeof(double) * 3 * cfg->n_u));
  HIPCHK(c, hipMalloc((void **)&c->dUcost, sizeof(double) * cfg->n_u));
  HIPCHK(c, hipMemcpyAsync(c->dU, c->U.data(), sizeof(double) * 3 * cfg->n_u, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->dUcost, ucost.data(), sizeof(double) * cfg->n_u, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->have_cfg = true;
  c->cfg_epoch++;
  return MPLX_OK;
}

extern "C" int mplx_set_capacity(mplx_ctx *c, int32_t n_slots, uint64_t max_nodes, uint64_t max_edges, uint64_t max_log) {
  if (!c) return MPLX_ERR_ARG;
  c->cfg_epoch++;
  if (n_slots > 0) c->n_slots = n_slots;
  if (max_nodes) c->cap_nodes = max_nodes;
  if (max_edges) c->cap_edges = max_edges;
  if (max_log) c->cap_log = max_log;
  return MPLX_OK;
}
extern "C" int mplx_set_pool_recycling(mplx_ctx *c, int32_t on) {
  if (!c) return MPLX_ERR_ARG;
  MPLX_REFUSE_PENDING(c);
  c->cfg_epoch++;
  c->recycle = on != 0;
  return MPLX_OK;
}
extern "C" int mplx_set_bucket_width(mplx_ctx *c, double w) {
  if (!c || w < 0) return MPLX_ERR_ARG;
  c->cfg_epoch++;
  c->bucket_width = w;
  return MPLX_OK;
}
extern "C" int mplx_set_speculation(mplx_ctx *c, int32_t mode) {
  if (!c) return MPLX_ERR_ARG;
  c->cfg_epoch++;
  c->speculation = mode;
  return MPLX_OK;
}
extern "C" int mplx_set_helpers(mplx_ctx *c, int32_t per_leader, int32_t reserved, uint64_t cache_rows) {
  if (!c || !(per_leader == -1 || per_leader == 0 || (per_leader >= 2 && per_leader <= 4))) return fail(c, MPLX_ERR_ARG, "helpers per leader: -1 (auto), 0 (off) or 2..4");
  MPLX_REFUSE_PENDING(c);  // (pools_valid = false below would re-allocate under the batch in flight)
  c->cfg_epoch++;
  c->helpers = per_leader;
  c->help_reserved = reserved;
  c->help_rows = cache_rows;
  c->pools_valid = false;  // the cache arrays are (re)sized with the pools
  return MPLX_OK;
}
extern "C" int mplx_helper_stats(const mplx_ctx *cc, uint32_t stats[4]) {
  if (!cc || !stats) return MPLX_ERR_ARG;
  mplx_ctx *c = const_cast<mplx_ctx *>(cc);
  if (c->help_stats_pending) {  // the read-back of the last batch completed with its stream synchronisation
    c->help_stats[0] = c->help_ctr_back[0];
    c->help_stats[1] = c->help_ctr_back[32];
    c->help_stats[2] = c->help_ctr_back[2];
    c->help_stats[3] = c->help_ctr_back[3];
    c->help_stats_pending = false;
#ifdef MPLX_HELP_DEBUG
    {
      fprintf(stderr, "[help debug] stall quits by (helper XCD row, leader XCD column) | attachments:\n");
      for (int r = 0; r < 8; r++) {
        for (int k = 0; k < 8; k++) fprintf(stderr, "%4u", c->help_ctr_back[128 + r * 8 + k]);
        fprintf(stderr, "   |");
        for (int k = 0; k < 8; k++) fprintf(stderr, "%5u", c->help_ctr_back[192 + r * 8 + k]);
        fprintf(stderr, "\n");
      }
      std::vector<HelpBox> hb(c->pool_slots);
      if (c->dbg_boxes && hipMemcpy(hb.data(), c->dbg_boxes, sizeof(HelpBox) * hb.size(), hipMemcpyDeviceToHost) == hipSuccess)
        for (size_t i = 0; i < hb.size(); i++)
          if (hb[i].pad1[0] > 1000000ull)
            fprintf(stderr, "[help debug] slot %zu (XCD %u): longest batch %.1f ms at %.1f ms into query %llu (batch %llu), helped %llu\n", i, (uint32_t)hb[i].xcc_plus1 - 1u,
                    hb[i].pad1[0] * 1e-5, (hb[i].pad1[1] >> 1) * 1e-5, hb[i].pad1[2], hb[i].pad1[3], hb[i].pad1[1] & 1ull);
    }
#endif
  }
  for (int i = 0; i < 4; i++) stats[i] = c->help_stats[i];
  return MPLX_OK;
}
extern "C" int mplx_set_record(mplx_ctx *c, uint32_t cap) {
  if (!c) return MPLX_ERR_ARG;
  c->cap_rec = cap;
  return MPLX_OK;
}

static void fill_params(const mplx_ctx *c, SearchParams &P, bool spec_kernels = false) {
  const mplx_config &g = c->cfg;
  P.control = g.control;
  P.n_u = g.n_u;
  P.ns = P.nk = state_len(g.control);
  P.dt = g.dt; P.v_max = g.v_max; P.a_max = g.a_max; P.j_max = g.j_max;
  P.w = g.w; P.eps = g.eps;
  P.tol_pos = g.tol_pos; P.tol_vel = g.tol_vel; P.tol_acc = g.tol_acc;
  P.t_max = g.t_max;
  P.max_expand = g.max_expand;
  P.heur_ignore_dynamics = g.heur_ignore_dynamics;
  P.U = c->dU;
  P.ucost = c->dUcost;
  P.map = map_dev(c);
  P.pot_weight = c->pot_weight;
  P.U_yaw = c->yaw ? c->dUyaw : nullptr;
  P.yaw_max = c->yaw ? g.yaw_max : 0.0;
  P.tol_yaw = c->yaw ? g.tol_yaw : -1.0;
  P.yaw_cos = 1.0;
  if (P.yaw_max > 0) {  // cos(yaw_max) by the same deterministic sequence the kernels evaluate the yaw direction with
    double sn;
    det_sincos(P.yaw_max, &sn, &P.yaw_cos);
  }
  // width of a coarse OPEN bucket in units of f: BUCKET_FACTOR edge costs of w dt (measurement: MPLX_BUCKET_FACTOR).  Deep searches
  // want narrow buckets (small near sets, no evictions), sparse OPEN lists wide ones (a pull per batch otherwise).  The kernels of
  // >= 256 threads pull a run of sparse buckets in one walk (mplx_kernels.h pull_fine_run), so the searches on the speculative
  // kernels take the narrow ones: 3 edge costs for the lattices of at most 64 inputs, half an edge cost for the larger ones (125-input
  // JRK: the OPEN list grows with the branching factor; C3 at 0.25 / 0.5 / 1 / 2 / 3: 7.06 / 7.00 / 7.24 | 7.08 / 7.35 / 7.41 s on two
  // boxes, the capped C4-JRK batch 654 / 656 / 656 ms at 1 / 2 / 3).  profiles/r06ae_merged_refill_sweep.txt: C4-ACC blocking step -4.9 %, its bulk phase
  // -3.2 %, C2 -16 %; profiles/r06m_bucket_width_sweep.txt: C3 -9 %.  The one-node kernels (64 / 128 lanes: no run pulls) keep 8.
  // The pop order does not depend on it.
  static const double bucket_factor_env = [] { const char *e = getenv("MPLX_BUCKET_FACTOR"); const double v = e ? atof(e) : 0.0; return v > 0 ? v : 0.0; }();
  const double bucket_factor = bucket_factor_env > 0 ? bucket_factor_env : (!spec_kernels ? 8.0 : g.n_u > 64 ? 0.5 : 3.0);
  P.bucket_width = c->bucket_width > 0 ? c->bucket_width : (g.w * g.dt > 0 ? g.w * g.dt * bucket_factor : 1.0);
  P.guard = c->guard;
}

// ------------------------------------------------------------------ launch guard, host side
// Every wait for a SEARCH launch goes through guard_wait(): poll the stream; past the context's deadline raise the abort
// word (the persistent loops of every search kernel read it and end their query with MPLX_PLAN_ABORTED), give the launch
// a grace period to leave, and return MPLX_ERR_TIMEOUT with the workgroups' watch records in the error text -- never
// block for ever (the reference's plan() always returns: mpl_test_node/src/map_planner_node.cpp:186-196).
static const char *guard_phase_name(uint32_t ph) {
  switch (ph) {
    case GUARD_BATCH: return "searching";
    case GUARD_CLAIM_WAIT: return "waiting for a table claim";
    case GUARD_ROW_WAIT: return "waiting for a look-ahead row";
    case GUARD_PROBE: return "table probe ran away";
    case GUARD_HELPER: return "helping";
    case GUARD_RECOVER: return "recoverTraj";
    case GUARD_PULL: return "far-bucket list closes on itself";
    case GUARD_DONE: return "left";
    case GUARD_TEST_HANG: return "test spin";
    case GUARD_START: return "query set-up";
    default: return "?";
  }
}
static std::string guard_dump(const mplx_ctx *c) {
  std::string out;
  int shown = 0, live = 0;
  for (int i = 0; i < GUARD_SLOTS; i++) {
    const unsigned long long w0 = __atomic_load_n(&c->guard->rec[i].w0, __ATOMIC_RELAXED), w1 = __atomic_load_n(&c->guard->rec[i].w1, __ATOMIC_RELAXED);
    if (!w0) continue;
    live++;
    if (shown >= 12) continue;
    char b[160];
    snprintf(b, sizeof(b), "%s[wg %d: %s, query %u, count %llu, info %llu]", shown ? " " : "", i, guard_phase_name((uint32_t)(w0 >> 56)), (unsigned)((w0 >> 40) & 0xFFFFu),
             w0 & 0xFFFFFFFFFFull, w1);
    out += b;
    shown++;
  }
  char b[64];
  snprintf(b, sizeof(b), " (%d workgroup records)", live);
  return out + b;
}
static void guard_arm(mplx_ctx *c) {  // before a search launch: no abort pending, no record of an earlier launch
  if (!c->guard) return;
  memset(c->guard, 0, sizeof(GuardBlock));
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
  c->guard_t0 = std::chrono::steady_clock::now();
}
// The clock starts at guard_arm() -- the launch -- so a submitted batch collected later and the two sequential waits of the
// moving-obstacle path (leaders, then helpers) share ONE deadline.  `keep_abort`: leave the abort word raised on return (the caller
// still has a second launch of the same search to drain -- mplx_poly_plan_batch's helpers -- and lowers it with guard_disarm()).
static int guard_wait(mplx_ctx *c, hipStream_t s, const char *what, bool keep_abort = false) {
  if (c->wedged) return fail(c, MPLX_ERR_TIMEOUT, "this context was lost to a launch that never ended (destroy it)");
  using clk = std::chrono::steady_clock;
  const auto t0 = c->guard_t0;
  const double limit = c->deadline_s, grace = 10.0;
  bool aborted = c->guard && __atomic_load_n(&c->guard->abort, __ATOMIC_RELAXED) != 0u;  // (raised by an earlier wait of the same launch)
  double t_abort = aborted ? std::chrono::duration<double>(clk::now() - t0).count() : 0.0;
  const auto tw = clk::now();
  std::string dump;
  for (;;) {
    const hipError_t e = hipStreamQuery(s);
    if (e == hipSuccess) break;
    if (e != hipErrorNotReady) return fail(c, MPLX_ERR_HIP, "hipStreamQuery failed while waiting for %s: %s", what, hipGetErrorString(e));
    const double el = std::chrono::duration<double>(clk::now() - t0).count();
    if (limit > 0 && !aborted && el > limit) {
      dump = guard_dump(c);
      __atomic_store_n(&c->guard->abort, 1u, __ATOMIC_SEQ_CST);
      aborted = true;
      t_abort = el;
    }
    if (aborted && el > t_abort + grace) {
      c->wedged = true;
      c->pending = false;
      return fail(c, MPLX_ERR_TIMEOUT, "%s did not end within %.1f s of its launch and did not answer the abort word for another %.0f s; the context is lost. At the deadline: %s; now: %s", what,
                  limit, grace, dump.c_str(), guard_dump(c).c_str());
    }
    const double waited = std::chrono::duration<double>(clk::now() - tw).count();
    if (waited < 0.002) sched_yield();
    else usleep(waited < 0.1 ? 50 : 250);
  }
  if (aborted) {
    if (!keep_abort) __atomic_store_n(&c->guard->abort, 0u, __ATOMIC_SEQ_CST);
    return fail(c, MPLX_ERR_TIMEOUT, "%s did not end within %.1f s of its launch and was aborted (results of the launch are void). At the deadline: %s", what, limit, dump.c_str());
  }
  return MPLX_OK;
}
static void guard_disarm(mplx_ctx *c) {
  if (c->guard) __atomic_store_n(&c->guard->abort, 0u, __ATOMIC_SEQ_CST);
}
extern "C" int mplx_set_deadline(mplx_ctx *c, double seconds) {
  if (!c) return MPLX_ERR_ARG;
  c->deadline_s = seconds;
  return MPLX_OK;
}
extern "C" int mplx_debug_hang_next_launch(mplx_ctx *c) {  // (tests) the next search launch spins until the deadline aborts it
  if (!c) return MPLX_ERR_ARG;
  c->debug_hang = true;
  return MPLX_OK;
}

// The shared state table before a batch launch: its slots carry the launch epoch (mplx_device.h tbl_empty / tbl_tagq), so the next
// batch just moves on to the next epoch -- what the previous ones left is "empty" to it -- and the table is cleared only when the
// epochs wrap (every 255th launch) or the table is new.  [Until round 5: a hipMemset of the whole table, 17-23 GB at C4 size, per batch.]
static int table_prepare(mplx_ctx *c, SearchParams &P, hipStream_t s) {
  static const bool always_clear = getenv("MPLX_TABLE_CLEAR") != nullptr;  // (diagnostics: the old behaviour)
  c->tbl_epoch++;
  if (c->tbl_epoch >= TBL_EPOCHS || always_clear) {
    HIPCHK(c, hipMemsetAsync(P.table, 0xFF, (size_t)(P.table_mask + 1) * sizeof(unsigned long long), s));
    c->tbl_epoch = 0;
  }
  P.tbl_epoch = c->tbl_epoch;
  return MPLX_OK;
}

template <typename T>
static int pool_alloc(mplx_ctx *c, T **p, size_t count) {
  void *v = nullptr;
  hipError_t e = hipMalloc(&v, count * sizeof(T));
  if (e != hipSuccess) {
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    return fail(c, MPLX_ERR_HIP, "hipMalloc(%zu bytes) failed: %s (device memory: %zu of %zu bytes free)", count * sizeof(T), hipGetErrorString(e), free_b, total_b);
  }
  c->pool_allocs.push_back(v);
  *p = (T *)v;
  return MPLX_OK;
}

static uint64_t next_pow2(uint64_t v) {
  uint64_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

static int ensure_pools(mplx_ctx *c, int slots) {
  const int control = c->cfg.control;
  if (c->pools_valid && c->pool_slots >= slots && c->pool_control == control && c->pool_nodes == c->cap_nodes &&
      c->pool_edges == c->cap_edges && c->pool_log == c->cap_log && (c->helpers != 0) == (c->pools.boxes != nullptr) && c->pool_recycle == c->recycle &&
      (c->helpers == 0 || c->pool_help_lanes == (c->cfg.n_u <= 31 ? 32 : 128)))
    return MPLX_OK;
  free_pools(c);
  SearchParams &P = c->pools;
  // every in-flight query needs at least one chunk of each pool
  uint64_t nch = (c->cap_nodes + (1u << NODE_CH_LOG) - 1) >> NODE_CH_LOG;
  uint64_t ech = (c->cap_edges + (1u << EDGE_CH_LOG) - 1) >> EDGE_CH_LOG;
  uint64_t och = (c->cap_log + (1u << OPEN_CH_LOG) - 1) >> OPEN_CH_LOG;
  if (nch < 1) nch = 1;
  if (ech < 1) ech = 1;
  if (och < 1) och = 1;
  // chunk ids are 16 bits in the kernels' LDS chunk tables: 2.1 G nodes / 4.3 G edges / 2.1 G log records
  if (nch > 0xFFFFull || ech > 0xFFFFull || och > 0xFFFFull) return fail(c, MPLX_ERR_ARG, "capacity too large (more than 65535 chunks in a pool)");
  if ((nch << NODE_CH_LOG) > (1ull << 30)) return fail(c, MPLX_ERR_ARG, "capacity too large (more than 2^30 states: the state table is indexed with 32 bits)");
  // Slots per pool state: 16 (load factor <= 0.06; 2^32 slots at most: a claimed slot's index travels in 32 bits).  The slowest of a
  // batch's ~190 look-up lanes sets the pace, and every extra probe is a dependent trip to memory: measured on one box (round 6,
  // profiles/r06j_ab_table_probe.json) the blocking C4-ACC step takes 3596 / 2498 / 2296 / 2192 ms with 2 / 4 / 8 / 16 slots per
  // state (recycled pools of 256 M states: the table also keeps an entry for every state of the batch's finished queries), the
  // capped query alone 2011 vs 1974 ms with 4 vs 16.  Rounds 1-5 used 4.  The table is cleared once per 255 launches, not per batch
  // (table_prepare), so its size costs memory only: 32 GB at C4 size.
  static const uint64_t fac_env = getenv("MPLX_TABLE_FACTOR") ? (uint64_t)atoi(getenv("MPLX_TABLE_FACTOR")) : 0;  // (measurement: slots per pool state)
  const uint64_t T = std::min<uint64_t>(next_pow2((fac_env ? fac_env : 16ull) * (nch << NODE_CH_LOG)), 1ull << 32);
  int r;
#define PA(ptr, cnt) if ((r = pool_alloc(c, &(ptr), (cnt))) != MPLX_OK) { free_pools(c); return r; }
  PA(P.node_pool, (size_t)(nch << NODE_CH_LOG) * rec_bytes(control));
  PA(P.edge_pool, (size_t)(ech << EDGE_CH_LOG) * EDGE_BYTES);
  PA(P.open_pool, (size_t)(och << OPEN_CH_LOG) * OPEN_BYTES);
  PA(P.table, (size_t)T);
  c->table_base = P.table;
  c->tbl_epoch = TBL_EPOCHS;  // (a fresh allocation: the first launch clears it and starts the epochs at 0 -- table_prepare)
  PA(P.bkt_head, (size_t)slots * 2 * NB * NSUB);
  HIPCHK(c, hipMemsetAsync(P.bkt_head, 0xFF, sizeof(uint32_t) * (size_t)slots * 2 * NB * NSUB, c->stream));  // all heads NIL; queries leave them so
  PA(P.chunk_next, 4);
  c->pool_recycle = c->recycle;
  c->d_chunk_bits = nullptr;
  if (c->recycle) {  // one bit per chunk, set = free: the image every launch starts from
    const uint64_t n[3] = {nch, ech, och};
    uint32_t w0 = 0;
    c->chunk_bits_init.clear();
    for (int k = 0; k < 3; k++) {
      c->chunk_word0[k] = w0;
      c->chunk_words[k] = (uint32_t)((n[k] + 31) / 32);
      for (uint32_t w = 0; w < c->chunk_words[k]; w++) {
        const uint64_t left = n[k] - 32ull * w;
        c->chunk_bits_init.push_back(left >= 32 ? 0xFFFFFFFFu : ((1u << left) - 1u));
      }
      w0 += c->chunk_words[k];
    }
    PA(c->d_chunk_bits, (size_t)w0);
  }
  P.boxes = nullptr; P.cache_c = nullptr; P.cache_h = nullptr; P.cache_next = nullptr; P.done_word = nullptr; P.cache_rows = 0;
  if (c->helpers != 0) {  // look-ahead cache of the helper workgroups (used by the speculative kernels, lattices <= 31 inputs)
    // rows of the heuristic cache: a quarter of the state capacity (a node is expanded ahead of time at most once),
    // at most 12 GiB unless the caller says otherwise; the row size follows the lattice (cache_row_doubles)
    const int unit_lanes = c->cfg.n_u <= 31 ? 32 : 128;
    const uint64_t row_bytes = (uint64_t)cache_row_doubles(unit_lanes) * sizeof(double);
    uint64_t rows = c->help_rows ? c->help_rows : std::max<uint64_t>((uint64_t)1 << 16, (nch << NODE_CH_LOG) / 4);
    if (!c->help_rows && rows * row_bytes > ((uint64_t)12 << 30)) rows = ((uint64_t)12 << 30) / row_bytes;
    if (rows > 0xFFFFFFF0ull) rows = 0xFFFFFFF0ull;
    PA(P.boxes, (size_t)slots + 1024);
    PA(P.cache_c, (size_t)(nch << NODE_CH_LOG));
    PA(P.cache_h, (size_t)rows * cache_row_doubles(unit_lanes));
    c->pool_help_lanes = unit_lanes;
    uint32_t *ctr = nullptr;
    PA(ctr, HELP_CTR_WORDS);                         // one 128-byte line per polled word
    P.cache_next = ctr;                              // [0] row counter, [2] [3] diagnostics
    P.done_word = (unsigned long long *)(ctr + 32);  // epoch << 32 | queries done
    P.cache_rows = (uint32_t)rows;
  }
#undef PA
  P.node_chunks = (uint32_t)nch;
  P.edge_chunks = (uint32_t)ech;
  P.open_chunks = (uint32_t)och;
  P.table_mask = T - 1;
  c->pool_slots = slots;
  c->pool_control = control;
  c->pool_nodes = c->cap_nodes;
  c->pool_edges = c->cap_edges;
  c->pool_log = c->cap_log;
  c->pools_valid = true;
  return MPLX_OK;
}

static int ensure_batch(mplx_ctx *c, int nq) {
  if (c->batch_cap >= nq && c->batch_rec == c->cap_rec) return MPLX_OK;
  free_batch(c);
  HIPCHK(c, hipMalloc((void **)&c->d_out, sizeof(QueryOut) * nq));
  HIPCHK(c, hipMalloc((void **)&c->d_in, sizeof(QueryIn) * nq));
  HIPCHK(c, hipMalloc((void **)&c->d_traj_nodes, sizeof(int32_t) * (size_t)nq * (MAX_TRAJ + 1)));
  HIPCHK(c, hipMalloc((void **)&c->d_traj_actions, sizeof(int32_t) * (size_t)nq * MAX_TRAJ));
  HIPCHK(c, hipMalloc((void **)&c->d_traj_states, sizeof(double) * (size_t)nq * (MAX_TRAJ + 1) * 13));
  HIPCHK(c, hipMalloc((void **)&c->d_traj_yaw, sizeof(double) * (size_t)nq * (MAX_TRAJ + 1)));
  HIPCHK(c, hipMalloc((void **)&c->d_next, sizeof(int32_t)));
  HIPCHK(c, hipMalloc((void **)&c->d_order, sizeof(int32_t) * nq));
  HIPCHK(c, hipMalloc((void **)&c->d_node_tables, sizeof(uint32_t) * (size_t)nq * MAX_NODE_CH));
  HIPCHK(c, hipMalloc((void **)&c->d_edge_tables, sizeof(uint32_t) * (size_t)nq * MAX_EDGE_CH));
  if (c->cap_rec) HIPCHK(c, hipMalloc((void **)&c->d_rec, sizeof(int32_t) * (size_t)nq * c->cap_rec));
  c->batch_cap = nq;
  c->batch_rec = c->cap_rec;
  return MPLX_OK;
}

static void wp_to_state(const mplx_waypoint &w, State &s) {
  for (int i = 0; i < 3; i++) {
    s.p[i] = w.pos[i];
    s.v[i] = (w.control & 2) ? w.vel[i] : 0.0;
    s.a[i] = (w.control & 4) ? w.acc[i] : 0.0;
    s.j[i] = (w.control & 8) ? w.jrk[i] : 0.0;
  }
}

static int pick_block(int n_u) { return n_u <= 64 ? 64 : (n_u <= 128 ? 128 : 256); }

template <int BLOCK>
static void launch_astar(int control, bool yaw, int grid, hipStream_t s, const SearchParams &P) {
  if (yaw) {
    switch (control) {
      case CTRL_VEL: hipLaunchKernelGGL((astar_kernel<BLOCK, CTRL_VEL, true>), dim3(grid), dim3(BLOCK), 0, s, P); break;
      case CTRL_ACC: hipLaunchKernelGGL((astar_kernel<BLOCK, CTRL_ACC, true>), dim3(grid), dim3(BLOCK), 0, s, P); break;
      case CTRL_JRK: hipLaunchKernelGGL((astar_kernel<BLOCK, CTRL_JRK, true>), dim3(grid), dim3(BLOCK), 0, s, P); break;
      default: hipLaunchKernelGGL((astar_kernel<BLOCK, CTRL_SNP, true>), dim3(grid), dim3(BLOCK), 0, s, P); break;
    }
    return;
  }
  switch (control) {
    case CTRL_VEL: hipLaunchKernelGGL((astar_kernel<BLOCK, CTRL_VEL>), dim3(grid), dim3(BLOCK), 0, s, P); break;
    case CTRL_ACC: hipLaunchKernelGGL((astar_kernel<BLOCK, CTRL_ACC>), dim3(grid), dim3(BLOCK), 0, s, P); break;
    case CTRL_JRK: hipLaunchKernelGGL((astar_kernel<BLOCK, CTRL_JRK>), dim3(grid), dim3(BLOCK), 0, s, P); break;
    default: hipLaunchKernelGGL((astar_kernel<BLOCK, CTRL_SNP>), dim3(grid), dim3(BLOCK), 0, s, P); break;
  }
}
template <int BLOCK>
static void launch_expand(int control, int grid, hipStream_t s, const SearchParams &P, const State *n, const double *t, int K, SuccOut *o, const double *yaw) {
  if (yaw) {
    switch (control) {
      case CTRL_VEL: hipLaunchKernelGGL((expand_kernel<BLOCK, CTRL_VEL, true>), dim3(grid), dim3(BLOCK), 0, s, P, n, t, K, o, yaw); break;
      case CTRL_ACC: hipLaunchKernelGGL((expand_kernel<BLOCK, CTRL_ACC, true>), dim3(grid), dim3(BLOCK), 0, s, P, n, t, K, o, yaw); break;
      case CTRL_JRK: hipLaunchKernelGGL((expand_kernel<BLOCK, CTRL_JRK, true>), dim3(grid), dim3(BLOCK), 0, s, P, n, t, K, o, yaw); break;
      default: hipLaunchKernelGGL((expand_kernel<BLOCK, CTRL_SNP, true>), dim3(grid), dim3(BLOCK), 0, s, P, n, t, K, o, yaw); break;
    }
    return;
  }
  switch (control) {
    case CTRL_VEL: hipLaunchKernelGGL((expand_kernel<BLOCK, CTRL_VEL>), dim3(grid), dim3(BLOCK), 0, s, P, n, t, K, o, nullptr); break;
    case CTRL_ACC: hipLaunchKernelGGL((expand_kernel<BLOCK, CTRL_ACC>), dim3(grid), dim3(BLOCK), 0, s, P, n, t, K, o, nullptr); break;
    case CTRL_JRK: hipLaunchKernelGGL((expand_kernel<BLOCK, CTRL_JRK>), dim3(grid), dim3(BLOCK), 0, s, P, n, t, K, o, nullptr); break;
    default: hipLaunchKernelGGL((expand_kernel<BLOCK, CTRL_SNP>), dim3(grid), dim3(BLOCK), 0, s, P, n, t, K, o, nullptr); break;
  }
}

// speculative kernels live in their own translation unit (mplx_spec_launch.hip) so the two halves of
// the device code compile in parallel; returns false when no variant fits (control kind / lattice size)
bool mplx_launch_spec(int speculation, int grid, hipStream_t s, const mplx::SearchParams &P);
// helper-assisted variant (mplx_help_launch.hip): leaders on s, helper workgroups on hs
bool mplx_launch_spec_help(int grid, hipStream_t s, const mplx::SearchParams &P);
// yaw-carrying states (mplx_yaw_launch.hip): ACC / JRK lattices of at most 128 inputs; false: none for the configuration
bool mplx_launch_spec_yaw(int grid, hipStream_t s, const mplx::SearchParams &P);
// FILTER builds (mplx_filter_launch.hip): SearchParams::filter_* decides which candidates are expanded; false: none for the configuration
bool mplx_launch_spec_filter(int grid, hipStream_t s, const mplx::SearchParams &P);

static int check_ready(mplx_ctx *c) {
  if (!c) return MPLX_ERR_ARG;
  if (!c->map) return fail(c, MPLX_ERR_ARG, "no map set (mplx_map_set)");
  if (!c->have_cfg) return fail(c, MPLX_ERR_ARG, "planner not configured (mplx_planner_config)");
  return MPLX_OK;
}

// ------------------------------------------------------------------ expand_batch
extern "C" int mplx_expand_batch(mplx_ctx *c, int K, const mplx_waypoint *nodes, mplx_succ *out) {
  int r = check_ready(c);
  if (r) return r;
  if (K <= 0 || !nodes || !out) return fail(c, MPLX_ERR_ARG, "bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  SearchParams P{};
  fill_params(c, P);
  std::vector<State> hs(K);
  std::vector<double> ht(K), hy(K);
  for (int i = 0; i < K; i++) {
    mplx_waypoint w = nodes[i];
    w.control = c->cfg.control;
    wp_to_state(w, hs[i]);
    ht[i] = nodes[i].t;
    hy[i] = nodes[i].yaw;
  }
  State *dn = nullptr;
  double *dt = nullptr, *dyaw = nullptr;
  SuccOut *dout = nullptr;
  const size_t no = (size_t)K * P.n_u;
  DevBufs bufs;
  HIPCHK(c, bufs.alloc(&dn, sizeof(State) * K));
  HIPCHK(c, bufs.alloc(&dt, sizeof(double) * K));
  HIPCHK(c, bufs.alloc(&dout, sizeof(SuccOut) * no));
  HIPCHK(c, hipMemcpyAsync(dn, hs.data(), sizeof(State) * K, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dt, ht.data(), sizeof(double) * K, hipMemcpyHostToDevice, c->stream));
  if (c->yaw) {
    HIPCHK(c, bufs.alloc(&dyaw, sizeof(double) * K));
    HIPCHK(c, hipMemcpyAsync(dyaw, hy.data(), sizeof(double) * K, hipMemcpyHostToDevice, c->stream));
  }
  const int grid = K < 4096 ? K : 4096;
  HIPCHK(c, hipEventRecord(c->ev0, c->stream));
  switch (pick_block(P.n_u)) {
    case 64: launch_expand<64>(P.control, grid, c->stream, P, dn, dt, K, dout, dyaw); break;
    case 128: launch_expand<128>(P.control, grid, c->stream, P, dn, dt, K, dout, dyaw); break;
    default: launch_expand<256>(P.control, grid, c->stream, P, dn, dt, K, dout, dyaw); break;
  }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(c->ev1, c->stream));
  HIPCHK(c, hipMemcpyAsync(out, dout, sizeof(SuccOut) * no, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipEventElapsedTime(&c->last_ms, c->ev0, c->ev1));
  return MPLX_OK;
}

extern "C" int mplx_heuristic_batch(mplx_ctx *c, int n, const mplx_waypoint *states, const mplx_waypoint *goal, double *h, int32_t *is_goal) {
  int r = check_ready(c);
  if (r) return r;
  if (n <= 0 || !states || !goal || !h || !is_goal) return fail(c, MPLX_ERR_ARG, "bad argument");
  if (!control_ok(goal->control)) return fail(c, MPLX_ERR_ARG, "bad goal control");
  HIPCHK(c, hipSetDevice(c->device));
  SearchParams P{};
  fill_params(c, P);
  HeurParams hp{};
  hp.w = P.w; hp.v_max = P.v_max; hp.heur_ignore_dynamics = P.heur_ignore_dynamics;
  hp.goal_control = goal->control;
  wp_to_state(*goal, hp.goal);
  hp.goal_nkey = state_key(goal->control, hp.goal, hp.goal_key);
  std::vector<State> hs(n);
  std::vector<double> ht(n);
  for (int i = 0; i < n; i++) {
    mplx_waypoint w = states[i];
    w.control = c->cfg.control;
    wp_to_state(w, hs[i]);
    ht[i] = states[i].t;
  }
  State *ds = nullptr;
  double *dt = nullptr, *dh = nullptr;
  int32_t *dg = nullptr;
  DevBufs bufs;
  HIPCHK(c, bufs.alloc(&ds, sizeof(State) * n));
  HIPCHK(c, bufs.alloc(&dt, sizeof(double) * n));
  HIPCHK(c, bufs.alloc(&dh, sizeof(double) * n));
  HIPCHK(c, bufs.alloc(&dg, sizeof(int32_t) * n));
  HIPCHK(c, hipMemcpyAsync(ds, hs.data(), sizeof(State) * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dt, ht.data(), sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(heuristic_kernel, dim3((n + 127) / 128), dim3(128), 0, c->stream, P, hp, n, ds, dt, dh, dg);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(h, dh, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(is_goal, dg, sizeof(int32_t) * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MPLX_OK;
}

// ------------------------------------------------------------------ plan
static void fill_result(const QueryOut &o, mplx_result &r) {
  r.status = o.status; r.traj_len = o.traj_len; r.cost = o.cost;
  r.n_expanded = o.n_expanded; r.n_closed = o.n_closed; r.n_nodes = o.n_nodes; r.n_edges = o.n_edges;
  r.n_primitives = o.n_primitives; r.n_succ = o.n_succ; r.n_succ_finite = o.n_succ_finite;
  r.voxel_reads = o.voxel_reads; r.n_push = o.n_push; r.n_reopen = o.n_reopen;
  r.n_refill = o.n_refill; r.n_evict = o.n_evict; r.expand_hash = o.expand_hash;
}

// First half of a batch: marshal the queries, reset the per-batch device state, launch the search on the context's stream.
// Nothing here waits for the device; the uploads' host-side sources stay alive in the context (pend_in / pend_order)
// until plan_batch_finish() has synchronised.
static int plan_batch_launch(mplx_ctx *c, int nq, const mplx_waypoint *starts, const mplx_waypoint *goals) {
  int r = check_ready(c);
  if (r) return r;
  if (nq <= 0 || !starts || !goals) return fail(c, MPLX_ERR_ARG, "bad argument");
  if (c->pending) return fail(c, MPLX_ERR_ARG, "a submitted batch is still outstanding on this context (mplx_plan_batch_wait first)");
  HIPCHK(c, hipSetDevice(c->device));
  const int slots = nq < c->n_slots ? nq : c->n_slots;
  if ((r = ensure_pools(c, slots)) != MPLX_OK) return r;
  if ((r = ensure_batch(c, nq)) != MPLX_OK) return r;
  std::vector<QueryIn> &in = c->pend_in;
  in.assign((size_t)nq, QueryIn{});
  for (int i = 0; i < nq; i++) {
    if (starts[i].enable_t) return fail(c, MPLX_ERR_ARG, "enable_t is not supported by the voxel-map environment");
    if (!control_ok(goals[i].control & ~MPLX_YAW)) return fail(c, MPLX_ERR_ARG, "bad goal control");
    mplx_waypoint s = starts[i];
    s.control = c->cfg.control;
    wp_to_state(s, in[i].start);
    wp_to_state(goals[i], in[i].goal);
    in[i].start_t = starts[i].t;
    in[i].goal_control = goals[i].control & ~MPLX_YAW;
    in[i].pad = 0;
    in[i].start_yaw = c->yaw ? starts[i].yaw : 0.0;
    in[i].goal_yaw = c->yaw ? goals[i].yaw : 0.0;
  }
  if (nq >= 0xFFFF) return fail(c, MPLX_ERR_ARG, "at most 65534 queries per batch");
  // launch order: longest expected search first (straight-line distance), so the tail of the batch
  // is made of short queries
  std::vector<int32_t> &order = c->pend_order;
  order.assign((size_t)nq, 0);
  {
    std::vector<std::pair<double, int32_t>> key(nq);
    for (int i = 0; i < nq; i++) {
      double d = 0;
      for (int k = 0; k < 3; k++) d += (starts[i].pos[k] - goals[i].pos[k]) * (starts[i].pos[k] - goals[i].pos[k]);
      key[i] = {-d, i};
    }
    std::stable_sort(key.begin(), key.end());
    for (int i = 0; i < nq; i++) order[i] = key[i].second;
  }
  SearchParams P = c->pools;
  // (the speculative kernels run ACC / JRK lattices of at most 128 inputs unless speculation is switched off: mplx_launch_spec*)
  fill_params(c, P, (c->speculation < 0 || c->speculation > 1) && (c->cfg.control == CTRL_ACC || c->cfg.control == CTRL_JRK) && c->cfg.n_u <= 128);
  if (c->wedged) return fail(c, MPLX_ERR_TIMEOUT, "this context was lost to a launch that never ended (destroy it)");
  const int xflags = getenv("MPLX_X_FLAGS") ? atoi(getenv("MPLX_X_FLAGS")) : 0;  // (diagnostics; read per launch so that a probe can switch between runs)
  P.xflags = xflags | (c->debug_hang ? 8 : 0);
  c->debug_hang = false;
  guard_arm(c);
  P.cap_rec = c->cap_rec;
  P.nq = nq;
  P.queries = c->d_in;
  P.order = c->d_order;
  P.out = c->d_out;
  P.traj_nodes = c->d_traj_nodes; P.traj_actions = c->d_traj_actions; P.traj_states = c->d_traj_states;
  P.traj_yaw = c->d_traj_yaw;
  P.rec_ids = c->cap_rec ? c->d_rec : nullptr;
  P.node_tables = c->d_node_tables;
  P.edge_tables = c->d_edge_tables;
  P.next_query = c->d_next;
  if (c->filter_table) filter_set(P, c->filter_table, c->filter_mask, c->filter_pool, c->filter_flag);
  HIPCHK(c, hipMemcpyAsync(c->d_order, order.data(), sizeof(int32_t) * nq, hipMemcpyHostToDevice, c->stream));
  if (int rt = table_prepare(c, P, c->stream)) return rt;
  HIPCHK(c, hipMemsetAsync(P.chunk_next, 0, 4 * sizeof(uint32_t), c->stream));
  P.chunk_bits = nullptr;
  HIPCHK(c, hipMemcpyAsync(c->d_in, in.data(), sizeof(QueryIn) * nq, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(c->d_next, 0, sizeof(int32_t), c->stream));
  // (yaw-carrying states run on the one-node kernels; a potential field / search region on the POT build of the
  //  speculative kernel for ACC lattices of at most 32 inputs -- mplx_launch_spec decides -- without helper workgroups)
  const bool spec = (c->speculation < 0 || c->speculation > 1) && !c->yaw;
  // pool recycling: the speculative kernels only (the one-node kernels bump-allocate), batches only (a single query has nobody to hand
  // its chunks to -- and its state space stays readable)
  const bool recycle = c->recycle && nq > 1 && spec && !c->filter_table && (P.control == CTRL_ACC || P.control == CTRL_JRK) && P.n_u <= 128 && c->d_chunk_bits;
  if (recycle) {
    HIPCHK(c, hipMemcpyAsync(c->d_chunk_bits, c->chunk_bits_init.data(), sizeof(uint32_t) * c->chunk_bits_init.size(), hipMemcpyHostToDevice, c->stream));
    P.chunk_bits = c->d_chunk_bits;
    for (int k = 0; k < 3; k++) { P.chunk_word0[k] = c->chunk_word0[k]; P.chunk_words[k] = c->chunk_words[k]; }
  }
  c->last_recycled = recycle;
  // Helper workgroups: a workgroup with no query (left) to lead expands the front of a running leader's OPEN list ahead
  // of time.  The launch holds one workgroup per compute unit at most, so all of them are resident together: for a
  // batch smaller than the machine the extra workgroups (blockIdx.x >= help_lead) help from the start; in a large
  // batch the leaders turn into helpers as they run out of queries.
  const int n_wg = c->n_cus;  // workgroups of the search kernel the machine holds at once (one per compute unit: 149 KB of LDS)
  int grid = slots;
  P.help_lead = slots;
  P.help_max = 0;
  P.help_limit = c->help_limit;
  const bool help = spec && !c->aux && !c->filter_table && (c->speculation < 0 || c->speculation >= 16) && c->helpers != 0 && P.boxes &&
                    ((P.n_u <= 31 && (P.control == CTRL_ACC || P.control == CTRL_JRK)) || (P.control == CTRL_JRK && P.n_u > 64 && P.n_u <= 128));
  if (help) {
    // auto: four helpers per leader for the lattices of at most 31 inputs (the capped query of the C4 batch alone: 1.95 s
    // with two, 1.89 s with four), two for the 65..128-input jerk lattices (no gain from more)
    P.help_max = c->helpers < 0 ? (P.n_u <= 31 ? 4 : 2) : c->helpers;
    P.help_lead = std::min(slots, n_wg);  // (the machine holds n_wg workgroups of these kernels: more would only wait)
    // a share of the machine that never leads: its workgroups help, from the start, the queries predicted longest
    // (the launch order is longest straight-line distance first; a batch lasts as long as its longest query).
    // auto: helpers for one sixteenth of the compute units' worth of leaders (16 leaders x helpers per leader on 256
    // compute units: 32 workgroups with two per leader, 64 with four) when the batch is at least twice the machine AND a
    // single query may run long (no expansion cap, or a cap of at least 200 000: with short capped queries every workgroup
    // is worth more leading -- the 125-input jerk batch capped at 20 000 loses 11 % to a reserved share).  Measured on the
    // C4-ACC batch (helpers per leader x reserved): 2 x 32: 2.190 s, 3 x 48: 2.166 s, 4 x 64: 2.151 s; 4 x 32 and 3 x 32
    // (fewer leaders covered: the capped query, 13th in the launch order, goes unhelped until the queue drains): 2.24 / 2.27 s
    const bool long_queries = P.max_expand <= 0 || P.max_expand >= 200000;
    const int reserved = c->help_reserved >= 0 ? c->help_reserved : (nq >= 2 * c->n_cus && long_queries ? std::max(1, c->n_cus / 16) * P.help_max : 0);
    if (reserved > 0 && slots + reserved > n_wg) P.help_lead = std::max(1, n_wg - reserved);
    grid = std::max(P.help_lead, std::min(P.help_lead * (P.help_max + 1), n_wg));
    if (const char *e = getenv("MPLX_HELP_GRID")) grid = std::max(P.help_lead, std::min(atoi(e), n_wg));  // (diagnostics: more would-be helpers than a leader takes)
    HIPCHK(c, hipMemsetAsync(P.boxes, 0, sizeof(HelpBox) * ((size_t)c->pool_slots + 1024), c->stream));
    c->dbg_boxes = P.boxes;
    // (diagnostic, tools/ab.py tail: MPLX_DEBUG_KEEP_CACHE=1 keeps the look-ahead cache of the previous launch -- the
    // same query planned again then finds an entry for every node, the fresh ones included: the time without any miss)
    static const bool keep_cache = getenv("MPLX_DEBUG_KEEP_CACHE") != nullptr;
    const bool kept = keep_cache && c->help_cache_filled;
    if (!kept) HIPCHK(c, hipMemsetAsync(P.cache_c, 0, sizeof(CacheRec) * ((size_t)P.node_chunks << NODE_CH_LOG), c->stream));
    c->help_cache_filled = true;
    // launch epoch: every word the helpers poll is tagged with it, so nothing left over from the previous launch
    // can be mistaken for progress of this one
    c->help_epoch++;
    if (c->help_epoch == 0) c->help_epoch = 1;
    P.epoch = c->help_epoch;
    c->help_done_init = (unsigned long long)P.epoch << 32;
    HIPCHK(c, hipMemsetAsync(P.cache_next, 0, HELP_CTR_WORDS * sizeof(uint32_t), c->stream));
    if (keep_cache && kept)  // the kept records name rows of the previous launch: new rows go behind them
      HIPCHK(c, hipMemcpyAsync(P.cache_next, &c->help_ctr_back[0], sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(P.done_word, &c->help_done_init, 8, hipMemcpyHostToDevice, c->stream));
  } else {
    P.boxes = nullptr;
  }
  HIPCHK(c, hipEventRecord(c->ev0, c->stream));
  bool launched = false;
  if (help) {
    launched = mplx_launch_spec_help(grid, c->stream, P);
  }
  if (!launched) {
    grid = slots;
    P.help_lead = grid;
  }
  // yaw-carrying states: the YAW build of the speculative kernel where one exists (ACC / JRK lattices of at most 128 inputs,
  // no auxiliary map), else the one-node kernel below
  bool launched_any = launched;
  if (c->filter_table) {  // (internal) the filtered search has its own builds and no fallback
    if (!mplx_launch_spec_filter(grid, c->stream, P)) return fail(c, MPLX_ERR_ARG, "no filtered build of the search kernel for this configuration");
    launched_any = true;
  }
  if (!launched_any && c->yaw && (c->speculation < 0 || c->speculation > 1)) launched_any = mplx_launch_spec_yaw(grid, c->stream, P);
  if (!launched_any && !(spec && mplx_launch_spec(c->speculation, grid, c->stream, P))) {
    switch (pick_block(P.n_u)) {
      case 64: launch_astar<64>(P.control, c->yaw, slots, c->stream, P); break;
      case 128: launch_astar<128>(P.control, c->yaw, slots, c->stream, P); break;
      default: launch_astar<256>(P.control, c->yaw, slots, c->stream, P); break;
    }
  }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(c->ev1, c->stream));
  c->pending = true;
  c->pend_nq = nq;
  c->pend_help = launched;
  return MPLX_OK;
}

// Second half: wait for the launch, read the results back.
static int plan_batch_finish(mplx_ctx *c, mplx_result *out) {
  if (!c->pending) return fail(c, MPLX_ERR_ARG, "no submitted batch on this context");
  HIPCHK(c, hipSetDevice(c->device));
  const int nq = c->pend_nq;
  c->pending = false;  // (whatever happens below, the batch is not outstanding any more)
  // Wait for the launch FIRST, with nothing else queued behind it: a device-to-host copy into pageable memory blocks its
  // caller until the stream has drained, which would put the host to sleep inside the very call the guard must watch.
  {
    const int rw = guard_wait(c, c->stream, "the search launch");
    if (rw) {
      c->last_nq = 0;  // nothing of an aborted launch is handed out
      return rw;
    }
  }
  if (c->pend_help) {
    HIPCHK(c, hipMemcpyAsync(c->help_ctr_back, c->pools.cache_next, HELP_CTR_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    c->help_stats_pending = true;
  } else {
    memset(c->help_stats, 0, sizeof(c->help_stats));
    c->help_stats_pending = false;
  }
  c->last_out.resize(nq);
  HIPCHK(c, hipMemcpyAsync(c->last_out.data(), c->d_out, sizeof(QueryOut) * nq, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));  // (copies only: the launch has ended)
  HIPCHK(c, hipEventElapsedTime(&c->last_ms, c->ev0, c->ev1));
  if (out)
    for (int i = 0; i < nq; i++) fill_result(c->last_out[i], out[i]);
  c->last_nq = nq;
  c->last_single = (nq == 1);
  c->last_control = c->cfg.control;
  c->last_dt = c->cfg.dt;
  c->last_U = c->U;
  c->last_yaw = c->yaw;
  c->last_Uyaw = c->Uyaw;
  c->last_map_epoch = c->map_epoch;
  c->plan_epoch++;
  return MPLX_OK;
}

extern "C" int mplx_plan_batch(mplx_ctx *c, int nq, const mplx_waypoint *starts, const mplx_waypoint *goals, mplx_result *out) {
  if (!out) return fail(c, MPLX_ERR_ARG, "bad argument");
  int r = plan_batch_launch(c, nq, starts, goals);
  if (r) return r;
  return plan_batch_finish(c, out);
}

// ---- streamed batches: the asynchronous pair of mplx_plan_batch.  submit() returns as soon as the batch is launched on
// the context's stream; wait() blocks until it has finished and hands out the results (mplx_result_traj etc. then answer
// for it).  One batch may be outstanding per context: several batches in flight = several contexts sharing one map
// replica (mplx_map_set_device on the same device pointer), which is what mplx_stream_* below packages.
extern "C" int mplx_plan_batch_submit(mplx_ctx *c, int nq, const mplx_waypoint *starts, const mplx_waypoint *goals) {
  return plan_batch_launch(c, nq, starts, goals);
}
extern "C" int mplx_plan_batch_wait(mplx_ctx *c, mplx_result *out) {
  if (!c) return MPLX_ERR_ARG;
  return plan_batch_finish(c, out);
}
extern "C" int mplx_plan_batch_done(mplx_ctx *c) {  // 1: wait() will not block; 0: still running; < 0: error
  if (!c) return MPLX_ERR_ARG;
  if (!c->pending) return 1;
  if (hipSetDevice(c->device) != hipSuccess) return MPLX_ERR_HIP;
  const hipError_t e = hipStreamQuery(c->stream);
  return e == hipSuccess ? 1 : e == hipErrorNotReady ? 0 : fail(c, MPLX_ERR_HIP, "hipStreamQuery failed: %s", hipGetErrorString(e));
}
extern "C" int mplx_set_helper_limit(mplx_ctx *c, int32_t limit) {
  if (!c) return MPLX_ERR_ARG;
  c->help_limit = limit < 0 ? -1 : limit;
  return MPLX_OK;
}
// give the pools back to the device allocator (they are re-created by the next plan): lets a caller hand the memory to
// other contexts -- a stream's lanes -- without destroying this one
extern "C" int mplx_release_pools(mplx_ctx *c) {
  if (!c) return MPLX_ERR_ARG;
  if (c->pending) return fail(c, MPLX_ERR_ARG, "a submitted batch is still outstanding on this context");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  free_pools(c);  // (the per-query result buffers stay: status / cost / trajectories of the last batch remain readable; the
                  //  state-space dumps -- mplx_result_nodes / _edges / _blocked -- need the pools and now fail loudly)
  return MPLX_OK;
}

// ---- mplx_stream: `depth` lanes (contexts of their own: stream, pools, result buffers) on the parent context's map replica
// and planner set-up.  submit() launches a batch on a free lane and returns a ticket; wait(ticket) collects it.  While
// the longest queries of batch n still run (a query is a serial pop chain: one compute unit plus its helpers), the
// workgroups of batch n + 1 take the compute units batch n no longer needs.  North star: "many independent start/goal
// queries ... shard one-query-per-stream"; the independence that licenses it: robot_team.hpp:60-66.
struct mplx_stream {
  mplx_ctx *parent = nullptr;
  std::vector<mplx_ctx *> lanes;
  std::vector<int64_t> ticket_of;   // ticket outstanding on each lane (-1: free)
  int64_t next_ticket = 0;
  uint64_t map_epoch = 0;           // the parent's map generation the lanes have adopted
  uint64_t cfg_epoch = 0;           // the parent's set-up generation the lanes have copied
  std::string err;
};
static int sfail(mplx_stream *s, int code, const char *msg) {
  if (s) s->err = msg ? msg : "";
  return code;
}
extern "C" const char *mplx_stream_last_error(const mplx_stream *s) { return s ? s->err.c_str() : ""; }
extern "C" void mplx_stream_destroy(mplx_stream *s) {
  if (!s) return;
  for (mplx_ctx *l : s->lanes) mplx_ctx_destroy(l);
  delete s;
}
// a lane takes the parent's planner set-up (control inputs, limits, tolerances, epsilon, cap) and, when the stream is
// created, its pool policy (mplx_stream_configure then owns that)
static int stream_lane_setup(mplx_ctx *parent, mplx_ctx *l, bool pools_too) {
  mplx_config cfg = parent->cfg;
  cfg.control = parent->cfg.control | (parent->yaw ? MPLX_YAW : 0);
  cfg.U = parent->U.data();
  cfg.U_yaw = parent->yaw ? parent->Uyaw.data() : nullptr;
  int r = mplx_planner_config(l, &cfg);
  if (r != MPLX_OK) return r;
  l->bucket_width = parent->bucket_width;
  l->speculation = parent->speculation;
  l->deadline_s = parent->deadline_s;
  l->recycle = parent->recycle;
  if (pools_too) {
    l->n_slots = parent->n_slots; l->cap_nodes = parent->cap_nodes; l->cap_edges = parent->cap_edges; l->cap_log = parent->cap_log;
    l->helpers = parent->helpers; l->help_reserved = parent->help_reserved; l->help_rows = parent->help_rows; l->help_limit = parent->help_limit;
  }
  return MPLX_OK;
}
extern "C" int mplx_stream_create(mplx_ctx *parent, int depth, mplx_stream **out) {
  if (!parent || !out || depth < 1 || depth > 8) return fail(parent, MPLX_ERR_ARG, "bad argument (depth 1..8)");
  *out = nullptr;
  int r = check_ready(parent);
  if (r) return r;
  mplx_stream *s = new mplx_stream();
  s->parent = parent;
  for (int k = 0; k < depth; k++) {
    mplx_ctx *l = nullptr;
    r = mplx_ctx_create(parent->device, &l);
    if (r == MPLX_OK) {
      s->lanes.push_back(l);
      r = mplx_map_set_device(l, parent->map, parent->dim, parent->origin, parent->res);  // the parent's replica, adopted: no copy
    }
    if (r == MPLX_OK) r = stream_lane_setup(parent, l, true);
    if (r != MPLX_OK) {
      if (l) parent->err = l->err.empty() ? g_create_error : l->err;
      mplx_stream_destroy(s);
      return r;
    }
  }
  s->ticket_of.assign((size_t)depth, -1);
  s->map_epoch = parent->map_epoch;
  s->cfg_epoch = parent->cfg_epoch;
  *out = s;
  return MPLX_OK;
}
// The lanes plan on the PARENT's map buffer.  When the parent's map changed since they adopted it (setMap / dilate / a
// VoxelGrid hand-over: another buffer, or other contents and another bitmap), every lane adopts it again -- which needs
// all of them idle: a batch in flight would be reading a buffer the parent may have freed.
static int stream_follow_map(mplx_stream *s) {
  if (s->parent->map_epoch == s->map_epoch) return MPLX_OK;
  for (mplx_ctx *l : s->lanes)
    if (l->pending) return sfail(s, MPLX_ERR_ARG, "the parent context's map changed while batches of the stream are in flight: wait for them before editing the map");
  for (mplx_ctx *l : s->lanes) {
    int r = mplx_map_set_device(l, s->parent->map, s->parent->dim, s->parent->origin, s->parent->res);
    if (r) return sfail(s, r, l->err.c_str());
  }
  s->map_epoch = s->parent->map_epoch;
  return MPLX_OK;
}
// per-lane knobs (the lanes copy the parent's at creation): pool capacities and helper policy of every lane
extern "C" int mplx_stream_configure(mplx_stream *s, int32_t n_slots, uint64_t total_nodes, uint64_t total_edges, uint64_t total_open_log,
                                     int32_t helpers_per_leader, int32_t helpers_reserved, uint64_t cache_rows, int32_t helper_limit) {
  if (!s) return MPLX_ERR_ARG;
  for (mplx_ctx *l : s->lanes) {
    if (l->pending) return sfail(s, MPLX_ERR_ARG, "a lane has a batch outstanding");
    mplx_set_capacity(l, n_slots, total_nodes, total_edges, total_open_log);
    int r = mplx_set_helpers(l, helpers_per_leader, helpers_reserved, cache_rows);
    if (r) return sfail(s, r, l->err.c_str());
    mplx_set_helper_limit(l, helper_limit);
  }
  return MPLX_OK;
}
extern "C" int mplx_stream_depth(const mplx_stream *s) { return s ? (int)s->lanes.size() : 0; }
extern "C" int mplx_stream_submit(mplx_stream *s, int nq, const mplx_waypoint *starts, const mplx_waypoint *goals, int64_t *ticket) {
  if (!s || !ticket) return MPLX_ERR_ARG;
  // A lane plans with the parent's set-up; what a lane cannot carry is refused, never silently dropped: the auxiliary map
  // (potential field / search region) belongs to the parent context and its POT kernels run without look-ahead helpers
  if (s->parent->aux) return sfail(s, MPLX_ERR_ARG, "the parent context has an auxiliary map (potential field / search region): streamed batches do not carry it -- use mplx_plan_batch");
  int rm = stream_follow_map(s);
  if (rm) return rm;
  if (s->parent->cfg_epoch != s->cfg_epoch) {  // the parent was re-configured since the lanes copied its set-up
    for (mplx_ctx *l : s->lanes)
      if (l->pending) return sfail(s, MPLX_ERR_ARG, "the parent context was re-configured while batches of the stream are in flight: wait for them first");
    for (mplx_ctx *l : s->lanes) {
      int r = stream_lane_setup(s->parent, l, false);
      if (r) return sfail(s, r, l->err.c_str());
    }
    s->cfg_epoch = s->parent->cfg_epoch;
  }
  for (size_t k = 0; k < s->lanes.size(); k++) {
    if (s->ticket_of[k] >= 0) continue;
    int r = plan_batch_launch(s->lanes[k], nq, starts, goals);
    if (r) return sfail(s, r, s->lanes[k]->err.c_str());
    s->ticket_of[k] = *ticket = s->next_ticket++;
    return MPLX_OK;
  }
  return sfail(s, MPLX_ERR_ARG, "every lane of the stream has a batch outstanding (mplx_stream_wait for one first)");
}
// the lane a ticket runs on (-1: unknown / already collected): its context answers mplx_result_traj etc. after wait()
static int stream_lane_of(const mplx_stream *s, int64_t ticket) {
  for (size_t k = 0; k < s->lanes.size(); k++)
    if (s->ticket_of[k] == ticket) return (int)k;
  return -1;
}
extern "C" int mplx_stream_done(mplx_stream *s, int64_t ticket) {
  if (!s) return MPLX_ERR_ARG;
  const int k = stream_lane_of(s, ticket);
  if (k < 0) return sfail(s, MPLX_ERR_ARG, "no such ticket");
  return mplx_plan_batch_done(s->lanes[(size_t)k]);
}
extern "C" int mplx_stream_wait(mplx_stream *s, int64_t ticket, mplx_result *out, mplx_ctx **lane_ctx) {
  if (!s) return MPLX_ERR_ARG;
  const int k = stream_lane_of(s, ticket);
  if (k < 0) return sfail(s, MPLX_ERR_ARG, "no such ticket");
  s->ticket_of[(size_t)k] = -1;
  int r = plan_batch_finish(s->lanes[(size_t)k], out);
  if (lane_ctx) *lane_ctx = s->lanes[(size_t)k];  // valid until the lane's next submit: trajectories, timings of this batch
  if (r) return sfail(s, r, s->lanes[(size_t)k]->err.c_str());
  return MPLX_OK;
}

extern "C" uint64_t mplx_plan_epoch(const mplx_ctx *c) { return c ? c->plan_epoch : 0; }

extern "C" const char *mplx_kernel_name(const mplx_ctx *c) {
  if (!c || !c->have_cfg) return "";
  const int control = c->cfg.control, n_u = c->cfg.n_u;
  const bool spec = (c->speculation < 0 || c->speculation > 1) && (control == CTRL_ACC || control == CTRL_JRK) && n_u <= 128 && !c->yaw;
  static thread_local char buf[64];
  const char *cn = control == CTRL_VEL ? "VEL" : control == CTRL_ACC ? "ACC" : control == CTRL_JRK ? "JRK" : "SNP";
  if (c->yaw && (c->speculation < 0 || c->speculation > 1) && (control == CTRL_ACC || control == CTRL_JRK) && n_u <= 128 && !c->aux) {
    snprintf(buf, sizeof(buf), "astar_spec_kernel<%d,%d,%s,yaw>", n_u <= 32 ? 32 : n_u <= 64 ? 64 : 128, n_u <= 32 ? 16 : 4, cn);
    return buf;
  }
  if (!spec) {
    snprintf(buf, sizeof(buf), c->yaw ? "astar_kernel<%d,%s,yaw>" : "astar_kernel<%d,%s>", pick_block(n_u), cn);
  } else {
    int ul, k;
    if (n_u <= 32 && c->speculation == 8) { ul = 64; k = 8; }
    else if (n_u <= 32) { ul = 32; k = 16; }
    else if (n_u <= 64) { ul = 64; k = 4; }
    else if (c->speculation == 2) { ul = 128; k = 2; }
    else { ul = 128; k = 4; }
    if (c->aux) { ul = n_u <= 32 ? 32 : 128; k = n_u <= 32 ? 16 : 4; }
    const bool help = !c->aux && (c->speculation < 0 || c->speculation >= 16) && c->helpers != 0 &&
                      ((ul == 32 && k == 16 && n_u <= 31) || (ul == 128 && k == 4 && control == CTRL_JRK && n_u > 64));
    snprintf(buf, sizeof(buf), c->aux ? "astar_spec_kernel<%d,%d,%s,pot>" :
                                help ? "astar_spec_kernel<%d,%d,%s,help>" : "astar_spec_kernel<%d,%d,%s>", ul, k, cn);
  }
  return buf;
}

extern "C" int mplx_plan(mplx_ctx *c, const mplx_waypoint *start, const mplx_waypoint *goal, mplx_result *out) {
  return mplx_plan_batch(c, 1, start, goal, out);
}

extern "C" int mplx_result_traj(mplx_ctx *c, int q, mplx_primitive *prs, mplx_waypoint *wps, int32_t *actions, int32_t *node_ids) {
  if (!c || q < 0 || q >= c->last_nq) return fail(c, MPLX_ERR_ARG, "no such query");
  MPLX_REFUSE_PENDING(c);  // (the batch in flight is overwriting the buffers the last batch's results live in)
  HIPCHK(c, hipSetDevice(c->device));
  const int len = c->last_out[q].traj_len;
  if (c->last_out[q].status != MPLX_PLAN_OK || len <= 0) return MPLX_OK;  // (MPLX_PLAN_TRAJ_TOO_LONG: cost only)
  std::vector<int32_t> tn(len + 1), ta(len);
  std::vector<double> ts((size_t)(len + 1) * 13), ty((size_t)len + 1, 0.0);
  if (c->last_yaw) HIPCHK(c, hipMemcpyAsync(ty.data(), c->d_traj_yaw + (size_t)q * (MAX_TRAJ + 1), sizeof(double) * (len + 1), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(tn.data(), c->d_traj_nodes + (size_t)q * (MAX_TRAJ + 1), sizeof(int32_t) * (len + 1), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(ta.data(), c->d_traj_actions + (size_t)q * MAX_TRAJ, sizeof(int32_t) * len, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(ts.data(), c->d_traj_states + (size_t)q * (MAX_TRAJ + 1) * 13, sizeof(double) * (len + 1) * 13, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // device order is goal -> start; emit start -> goal.  Control kind, dt and U are the ones the plan ran with.
  const int control = c->last_control, out_control = control | (c->last_yaw ? MPLX_YAW : 0);
  for (int i = 0; i <= len; i++) {
    const double *s = &ts[(size_t)(len - i) * 13];
    if (wps) {
      mplx_waypoint &w = wps[i];
      memset(&w, 0, sizeof(w));
      for (int k = 0; k < 3; k++) {
        w.pos[k] = s[k]; w.vel[k] = s[3 + k]; w.acc[k] = s[6 + k]; w.jrk[k] = s[9 + k];
      }
      w.t = s[12];
      w.yaw = ty[len - i];
      w.control = out_control;
    }
    if (node_ids) node_ids[i] = tn[len - i];
  }
  for (int i = 0; i < len; i++) {
    const int a = ta[len - 1 - i];
    if (actions) actions[i] = a;
    if (prs) {  // forward_action(parent coord, action): primitive from the stored parent state
      const double *s = &ts[(size_t)(len - i) * 13];
      mplx_primitive &p = prs[i];
      memset(&p, 0, sizeof(p));
      for (int ax = 0; ax < 3; ax++) prim_build_axis(control, s[ax], s[3 + ax], s[6 + ax], s[9 + ax], c->last_U[3 * a + ax], p.c[ax]);
      p.t = c->last_dt;
      p.control = out_control;
      if (c->last_yaw) {  // the VEL-type yaw channel from the parent's yaw
        p.cyaw[4] = c->last_Uyaw[a];
        p.cyaw[5] = ty[len - i];
      }
    }
  }
  return MPLX_OK;
}

extern "C" int mplx_result_expanded(mplx_ctx *c, int q, uint32_t cap, int32_t *ids, uint32_t *n) {
  if (!c || q < 0 || q >= c->last_nq || !ids || !n) return fail(c, MPLX_ERR_ARG, "bad argument");
  MPLX_REFUSE_PENDING(c);
  if (!c->batch_rec || !c->d_rec) return fail(c, MPLX_ERR_ARG, "recording disabled (mplx_set_record)");
  HIPCHK(c, hipSetDevice(c->device));
  uint32_t cnt = c->last_out[q].n_recorded;
  if (cnt > cap) cnt = cap;
  if (cnt) HIPCHK(c, hipMemcpyAsync(ids, c->d_rec + (size_t)q * c->batch_rec, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *n = cnt;
  return MPLX_OK;
}

// StateSpace predecessor lists of the last single plan: for every node in id order, its edges oldest first
extern "C" int mplx_result_edges(mplx_ctx *c, int32_t *child, int32_t *parent, int32_t *action, uint64_t cap, uint64_t *n_out) {
  if (!c || !c->last_single || !c->pools_valid || !n_out) return fail(c, MPLX_ERR_ARG, "predecessor dump needs a preceding single mplx_plan()");
  MPLX_REFUSE_PENDING(c);
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n_nodes = c->last_out[0].n_nodes, n_edges = c->last_out[0].n_edges;
  *n_out = n_edges;
  if (n_nodes == 0 || n_edges == 0 || cap == 0) return MPLX_OK;
  const int rb = rec_bytes(c->pool_control);
  std::vector<uint32_t> ntbl(MAX_NODE_CH), etbl(MAX_EDGE_CH);
  HIPCHK(c, hipMemcpyAsync(ntbl.data(), c->d_node_tables, sizeof(uint32_t) * MAX_NODE_CH, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(etbl.data(), c->d_edge_tables, sizeof(uint32_t) * MAX_EDGE_CH, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // predecessor heads of all nodes, then the query's edge records, chunk by chunk
  std::vector<uint32_t> head(n_nodes);
  {
    const size_t per = (size_t)1 << NODE_CH_LOG;
    std::vector<char> buf(per * rb);
    for (size_t base = 0; base < n_nodes; base += per) {
      const size_t cnt = n_nodes - base < per ? n_nodes - base : per;
      const uint32_t ch = ntbl[base >> NODE_CH_LOG];
      if (ch == NIL) return fail(c, MPLX_ERR_ARG, "inconsistent chunk table");
      HIPCHK(c, hipMemcpyAsync(buf.data(), c->pools.node_pool + ((size_t)ch << NODE_CH_LOG) * rb, cnt * rb, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      for (size_t k = 0; k < cnt; k++) memcpy(&head[base + k], buf.data() + k * rb + 20, 4);
    }
  }
  std::vector<EdgeRec> edges(n_edges);
  {
    const size_t per = (size_t)1 << EDGE_CH_LOG;
    for (size_t base = 0; base < n_edges; base += per) {
      const size_t cnt = n_edges - base < per ? n_edges - base : per;
      const uint32_t ch = etbl[base >> EDGE_CH_LOG];
      if (ch == NIL) return fail(c, MPLX_ERR_ARG, "inconsistent chunk table");
      HIPCHK(c, hipMemcpyAsync(edges.data() + base, c->pools.edge_pool + ((size_t)ch << EDGE_CH_LOG) * EDGE_BYTES, cnt * EDGE_BYTES, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  uint64_t w = 0;
  std::vector<uint32_t> lst;
  for (size_t i = 0; i < n_nodes; i++) {
    lst.clear();  // the device list is newest first; the reference's pred vectors grow by push_back
    for (uint32_t e = head[i]; e != NIL; e = edges[e].next) {
      if (e >= n_edges || lst.size() > n_edges) return fail(c, MPLX_ERR_ARG, "corrupt predecessor list of node %zu", i);
      lst.push_back(e);
    }
    for (size_t k = lst.size(); k-- > 0;) {
      const uint32_t e = lst[k];
      if (w < cap) {
        if (child) child[w] = (int32_t)i;
        if (parent) parent[w] = (int32_t)edges[e].parent;
        if (action) action[w] = (int32_t)(edges[e].action & EDGE_ACTION_MASK);
      }
      w++;
    }
  }
  *n_out = w;
  return MPLX_OK;
}

extern "C" int mplx_result_nodes(mplx_ctx *c, uint64_t cap, mplx_waypoint *coords, double *g, double *h, int32_t *closed, int32_t *opened) {
  if (!c || !c->last_single || !c->pools_valid) return fail(c, MPLX_ERR_ARG, "state-space dump needs a preceding single mplx_plan()");
  MPLX_REFUSE_PENDING(c);
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = c->last_out[0].n_nodes;
  if (n == 0) return MPLX_OK;
  if ((uint64_t)n > cap) return fail(c, MPLX_ERR_CAPACITY, "state-space dump: the last plan created %zu states, the caller's arrays hold %llu", n, (unsigned long long)cap);
  const int control = c->pool_control, nk = state_len(control), rb = rec_bytes(control), hot = rec_hot_bytes(control);
  std::vector<uint32_t> tbl(MAX_NODE_CH);
  HIPCHK(c, hipMemcpyAsync(tbl.data(), c->d_node_tables, sizeof(uint32_t) * MAX_NODE_CH, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const size_t per = (size_t)1 << NODE_CH_LOG;
  std::vector<char> buf(per * rb);
  for (size_t base = 0; base < n; base += per) {
    const size_t cnt = n - base < per ? n - base : per;
    const uint32_t ch = tbl[base >> NODE_CH_LOG];
    if (ch == NIL) return fail(c, MPLX_ERR_ARG, "inconsistent chunk table");
    HIPCHK(c, hipMemcpyAsync(buf.data(), c->pools.node_pool + ((size_t)ch << NODE_CH_LOG) * rb, cnt * rb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (size_t k = 0; k < cnt; k++) {
      const char *r = buf.data() + k * rb;
      const size_t i = base + k;
      uint32_t fl;
      memcpy(&fl, r + 16, 4);
      if (g) memcpy(&g[i], r, 8);
      if (h) memcpy(&h[i], r + 8, 8);
      if (closed) closed[i] = (fl & FLAG_CLOSED) ? 1 : 0;
      if (opened) opened[i] = (fl & FLAG_OPENED) ? 1 : 0;
      if (coords) {
        mplx_waypoint &w = coords[i];
        memset(&w, 0, sizeof(w));
        const double *st = (const double *)(r + hot);
        for (int d = 0; d < nk; d++) {
          double *dst = d < 3 ? w.pos : d < 6 ? w.vel : d < 9 ? w.acc : w.jrk;
          dst[d % 3] = st[d];
        }
        w.t = st[nk + (c->last_yaw ? 1 : 0)];
        if (c->last_yaw) w.yaw = st[nk];
        w.control = control | (c->last_yaw ? MPLX_YAW : 0);
      }
    }
  }
  return MPLX_OK;
}

// (diagnostics) the raw node records of query q of the last batch -- {g, h, flags, pred, key[], state...} as the kernels
// keep them (mplx_device.h) -- for offline consistency checks: keys against states, duplicate keys.  *rec_size: bytes per
// record; bytes: room for *n_records x *rec_size (MPLX_ERR_CAPACITY, with the counts filled in, when cap_bytes is too small).
extern "C" int mplx_debug_query_records(mplx_ctx *c, int q, uint64_t cap_bytes, void *bytes, uint64_t *n_records, int32_t *rec_size) {
  if (!c || q < 0 || q >= c->last_nq || !c->pools_valid || !n_records || !rec_size) return fail(c, MPLX_ERR_ARG, "bad argument / no batch / pools released");
  if (c->last_recycled) return fail(c, MPLX_ERR_ARG, "the last batch ran with pool recycling (mplx_set_pool_recycling): the state spaces of its queries were handed back to the pools");
  MPLX_REFUSE_PENDING(c);
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = c->last_out[q].n_nodes;
  const int rb = rec_bytes(c->pool_control);
  *n_records = n;
  *rec_size = rb;
  if ((uint64_t)n * rb > cap_bytes || !bytes) return fail(c, MPLX_ERR_CAPACITY, "record dump: %zu records of %d bytes", n, rb);
  std::vector<uint32_t> tbl(MAX_NODE_CH);
  HIPCHK(c, hipMemcpyAsync(tbl.data(), c->d_node_tables + (size_t)q * MAX_NODE_CH, sizeof(uint32_t) * MAX_NODE_CH, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const size_t per = (size_t)1 << NODE_CH_LOG;
  for (size_t base = 0; base < n; base += per) {
    const size_t cnt = n - base < per ? n - base : per;
    const uint32_t ch = tbl[base >> NODE_CH_LOG];
    if (ch == NIL) return fail(c, MPLX_ERR_ARG, "inconsistent chunk table");
    HIPCHK(c, hipMemcpyAsync((char *)bytes + base * rb, c->pools.node_pool + ((size_t)ch << NODE_CH_LOG) * rb, cnt * rb, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MPLX_OK;
}

// Successors with cost +inf of the LAST single mplx_plan(): the reference's GraphSearch gives every successor
// get_succ returns an hm_ entry and a pred_* entry, also the blocked ones (cost inf, env_poly_map.h:60-66); the
// device search keeps only what can be relaxed.  get_succ is a pure function of (node, U, dt, limits, map), so the
// blocked primitives are re-derived here on request -- one expand launch over the closed nodes -- instead of being
// stored on the hot path.  parent / action: one entry per blocked primitive, parents in node-id order, actions
// ascending; *n_states_all = hm_.size() as upstream counts it (states reached with finite cost + states that only
// blocked primitives reach).  Needs the map and the planner set-up of that plan to be still in place.
extern "C" int mplx_result_blocked(mplx_ctx *c, int32_t *parent, int32_t *action, uint64_t cap, uint64_t *n_out, uint64_t *n_states_all) {
  if (!c || !c->last_single || !c->pools_valid || !n_out) return fail(c, MPLX_ERR_ARG, "blocked-primitive dump needs a preceding single mplx_plan()");
  MPLX_REFUSE_PENDING(c);
  if (c->last_control != c->cfg.control || c->last_dt != c->cfg.dt || c->last_U != c->U || c->last_yaw != c->yaw || c->last_Uyaw != c->Uyaw)
    return fail(c, MPLX_ERR_ARG, "the planner was re-configured since the plan: blocked primitives cannot be re-derived");
  if (c->last_map_epoch != c->map_epoch)
    return fail(c, MPLX_ERR_ARG, "the map changed since the plan: its blocked primitives cannot be re-derived (they would be computed against the new map)");
  const float plan_ms = c->last_ms;  // the internal expand launches below must not replace the plan's kernel time
  struct RestoreMs { mplx_ctx *c; float ms; ~RestoreMs() { c->last_ms = ms; } } restore_ms{c, plan_ms};
  const size_t n = c->last_out[0].n_nodes;
  *n_out = 0;
  if (n_states_all) *n_states_all = n;
  if (n == 0) return MPLX_OK;
  std::vector<mplx_waypoint> coords(n);
  std::vector<int32_t> closed(n);
  int r = mplx_result_nodes(c, n, coords.data(), nullptr, nullptr, closed.data(), nullptr);
  if (r) return r;
  const int control = c->last_control, nk = state_len(control), n_u = c->cfg.n_u;
  auto key_of = [&](const mplx_waypoint &w) {
    State st;
    mplx_waypoint ww = w;
    ww.control = control;
    wp_to_state(ww, st);
    int32_t k[MAX_KEY];
    state_key(control, st, k);
    std::string s((const char *)k, sizeof(int32_t) * (size_t)nk);
    if (c->last_yaw) {
      const int32_t yk = (int32_t)round(w.yaw / KEY_RES_YAW);
      s.append((const char *)&yk, sizeof(yk));
    }
    return s;
  };
  std::vector<std::string> keys(n);
  for (size_t i = 0; i < n; i++) keys[i] = key_of(coords[i]);
  std::vector<std::string> sorted_keys = keys;
  std::sort(sorted_keys.begin(), sorted_keys.end());
  std::vector<int32_t> ids;
  for (size_t i = 0; i < n; i++)
    if (closed[i]) ids.push_back((int32_t)i);
  std::vector<std::string> only_blocked;
  uint64_t w = 0;
  const size_t CH = 8192;
  std::vector<mplx_waypoint> batch;
  std::vector<mplx_succ> succ;
  for (size_t base = 0; base < ids.size(); base += CH) {
    const size_t cnt = std::min(CH, ids.size() - base);
    batch.resize(cnt);
    for (size_t k = 0; k < cnt; k++) batch[k] = coords[ids[base + k]];
    succ.resize(cnt * (size_t)n_u);
    r = mplx_expand_batch(c, (int)cnt, batch.data(), succ.data());
    if (r) return r;
    for (size_t k = 0; k < cnt; k++)
      for (int a = 0; a < n_u; a++) {
        const mplx_succ &sc = succ[k * (size_t)n_u + a];
        if (!sc.valid || !std::isinf(sc.cost)) continue;
        if (w < cap) {
          if (parent) parent[w] = ids[base + k];
          if (action) action[w] = a;
        }
        w++;
        std::string ks = c->last_yaw ? key_of(sc.wp) : std::string((const char *)sc.key, sizeof(int32_t) * (size_t)nk);
        if (!std::binary_search(sorted_keys.begin(), sorted_keys.end(), ks)) only_blocked.push_back(ks);
      }
  }
  std::sort(only_blocked.begin(), only_blocked.end());
  only_blocked.erase(std::unique(only_blocked.begin(), only_blocked.end()), only_blocked.end());
  *n_out = w;
  if (n_states_all) *n_states_all = n + only_blocked.size();
  return MPLX_OK;
}

extern "C" int mplx_result_timing(mplx_ctx *c, int q, double *t_begin_s, double *t_end_s, int32_t *slot) {
  if (!c || q < 0 || q >= c->last_nq) return fail(c, MPLX_ERR_ARG, "no such query");
  // wall_clock64() ticks at 100 MHz on gfx950; times are relative to the first query start of the batch
  unsigned long long t0 = ~0ull;
  for (int i = 0; i < c->last_nq; i++)
    if (c->last_out[i].t_begin && c->last_out[i].t_begin < t0) t0 = c->last_out[i].t_begin;
  if (t_begin_s) *t_begin_s = (double)(c->last_out[q].t_begin - t0) * 1e-8;
  if (t_end_s) *t_end_s = (double)(c->last_out[q].t_end - t0) * 1e-8;
  if (slot) *slot = (int32_t)c->last_out[q].slot;
  return MPLX_OK;
}

extern "C" int mplx_result_cycles(mplx_ctx *c, int q, uint64_t cyc[10]) {
  if (!c || q < 0 || q >= c->last_nq || !cyc) return fail(c, MPLX_ERR_ARG, "no such query");
  for (int i = 0; i < 10; i++) cyc[i] = c->last_out[q].cyc[i];
  return MPLX_OK;
}

extern "C" int mplx_result_speculation(mplx_ctx *c, int q, uint64_t spec[4]) {
  if (!c || q < 0 || q >= c->last_nq || !spec) return fail(c, MPLX_ERR_ARG, "no such query");
  for (int i = 0; i < 4; i++) spec[i] = c->last_out[q].spec[i];
  return MPLX_OK;
}

extern "C" int mplx_last_kernel_ms(const mplx_ctx *c, float *ms) {
  if (!c || !ms) return MPLX_ERR_ARG;
  *ms = c->last_ms;
  return MPLX_OK;
}
#include "mplx_lpa.inl"
#include "mplx_grid.inl"
#include "mplx_poly_search.h"
#include "mplx_poly.inl"
