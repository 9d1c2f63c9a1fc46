// mplx_grid.inl -- device-resident VoxelGrid (the map-ingest step in front of the search).
// Included at the end of mplx_api.hip (same translation unit: shares the utility kernels).
//
// Mirrors the in-tree planning_ros_utils/src/mapping_utils/voxel_grid.cpp: two int8 grids (map_,
// inflated_map_), float resolution, truncating floatToInt (:201-203).  The grids live in HBM in the
// VoxelMap.data layout (x fastest, voxel_grid.cpp:88) instead of boost::multi_array's [x][y][z], so
// getMap() is a value transform and mplx_grid_to_map() hands the result to a planner context device
// to device -- a replanning cycle (clear / fill / addCloud -> getMap -> setMap -> plan,
// map_replanner_node.cpp:175-230) never crosses PCIe.

namespace mplx {

struct GridDev {
  int8_t *map, *inflated;
  int32_t dim[3];
  double origin_d[3];
  float res;
};

__device__ __forceinline__ bool grid_cell_of(const GridDev &g, const double *pt, int32_t *pn) {
  bool in = true;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    pn[i] = (int32_t)((pt[i] - g.origin_d[i]) / (double)g.res);  // .cast<int>(): towards zero
    in = in && pn[i] >= 0 && pn[i] < g.dim[i];
  }
  return in;
}
__device__ __forceinline__ size_t grid_idx(const GridDev &g, int x, int y, int z) { return (size_t)x + (size_t)g.dim[0] * y + (size_t)g.dim[0] * g.dim[1] * z; }

__global__ void grid_add_cloud_kernel(GridDev g, int n, const double *pts) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int32_t pn[3];
  if (grid_cell_of(g, pts + 3 * (size_t)i, pn)) g.map[grid_idx(g, pn[0], pn[1], pn[2])] = 100;
}
// addCloud(pts, ns), voxel_grid.cpp:191-207, is a sequential loop: a point inflates its neighbourhood only
// if its cell is not occupied yet (by the grid or by an EARLIER point of the call), and new_obs lists the
// inflated cells in the order they flip.  Parallel form: (A) first[cell] = smallest point index landing in
// it; (B) the first point of a cell that was not occupied offers key = i * n_ns + k to every neighbour
// that is not inflated yet, keymin[cell] = smallest key; (C) the winners emit (key, cell), the host sorts
// by key -- the sequential order; all points mark map_.  The scratch arrays are restored on the way out.
__global__ void grid_ns_first_kernel(GridDev g, int n, const double *pts, uint32_t *first) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int32_t pn[3];
  if (grid_cell_of(g, pts + 3 * (size_t)i, pn)) atomicMin(&first[grid_idx(g, pn[0], pn[1], pn[2])], (uint32_t)i);
}
__global__ void grid_ns_offer_kernel(GridDev g, int n, const double *pts, int n_ns, const int32_t *ns, const uint32_t *first, unsigned long long *keymin) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * n_ns) return;
  const int i = (int)(t / n_ns), k = (int)(t % n_ns);
  int32_t pn[3];
  if (!grid_cell_of(g, pts + 3 * (size_t)i, pn)) return;
  const size_t c = grid_idx(g, pn[0], pn[1], pn[2]);
  if (first[c] != (uint32_t)i || g.map[c] == 100) return;
  const int x = pn[0] + ns[3 * k], y = pn[1] + ns[3 * k + 1], z = pn[2] + ns[3 * k + 2];
  if (x < 0 || x >= g.dim[0] || y < 0 || y >= g.dim[1] || z < 0 || z >= g.dim[2]) return;
  const size_t c2 = grid_idx(g, x, y, z);
  if (g.inflated[c2] != 100) atomicMin(&keymin[c2], (unsigned long long)t);
}
__global__ void grid_ns_emit_kernel(GridDev g, int n, const double *pts, int n_ns, const int32_t *ns, const uint32_t *first, const unsigned long long *keymin,
                                    unsigned long long *out_key, int32_t *out_cell, unsigned int *out_n) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * n_ns) return;
  const int i = (int)(t / n_ns), k = (int)(t % n_ns);
  int32_t pn[3];
  if (!grid_cell_of(g, pts + 3 * (size_t)i, pn)) return;
  const size_t c = grid_idx(g, pn[0], pn[1], pn[2]);
  if (first[c] != (uint32_t)i || g.map[c] == 100) return;
  const int x = pn[0] + ns[3 * k], y = pn[1] + ns[3 * k + 1], z = pn[2] + ns[3 * k + 2];
  if (x < 0 || x >= g.dim[0] || y < 0 || y >= g.dim[1] || z < 0 || z >= g.dim[2]) return;
  const size_t c2 = grid_idx(g, x, y, z);
  if (keymin[c2] == (unsigned long long)t) {
    const unsigned int o = atomicAdd(out_n, 1u);
    out_key[o] = (unsigned long long)t;
    out_cell[3 * (size_t)o] = x; out_cell[3 * (size_t)o + 1] = y; out_cell[3 * (size_t)o + 2] = z;
  }
}
// apply: winners' cells become inflated, every point's cell becomes occupied, scratch is restored
__global__ void grid_ns_apply_kernel(GridDev g, int n, const double *pts, int n_ns, const int32_t *ns, uint32_t *first, unsigned long long *keymin,
                                     const int32_t *out_cell, unsigned int n_out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < (long long)n_out) {
    const size_t c2 = grid_idx(g, out_cell[3 * t], out_cell[3 * t + 1], out_cell[3 * t + 2]);
    g.inflated[c2] = 100;
    keymin[c2] = ~0ull;
  }
  if (t < (long long)n) {
    int32_t pn[3];
    if (grid_cell_of(g, pts + 3 * (size_t)t, pn)) {
      const size_t c = grid_idx(g, pn[0], pn[1], pn[2]);
      g.map[c] = 100;
      first[c] = 0xFFFFFFFFu;
    }
  }
}
__global__ void grid_decay_kernel(int8_t *a, int8_t *b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int8_t v = a[i];
    if (v > 0) a[i] = v - 1;
    v = b[i];
    if (v > 0) b[i] = v - 1;
  }
}
__global__ void grid_column_kernel(GridDev g, int nx, int ny, int8_t val) {
  for (int z = blockIdx.x * blockDim.x + threadIdx.x; z < g.dim[2]; z += gridDim.x * blockDim.x) g.map[grid_idx(g, nx, ny, z)] = val;
}
// getMap / getInflatedMap (voxel_grid.cpp:71-127): > 0 -> 100, everything else (free, unknown) -> 0
__global__ void grid_get_map_kernel(const int8_t *in, int8_t *out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i] > 0 ? 100 : 0;
}
// allocate (voxel_grid.cpp:129-181): the overlap of the old grid is carried into the new one
__global__ void grid_realloc_kernel(const int8_t *old, int odx, int ody, int odz, int oox, int ooy, int ooz, int8_t *nw, int ndx, int ndy, int ndz, int nox,
                                    int noy, int noz) {
  const size_t n = (size_t)ndx * ndy * ndz;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int l = (int)(i % ndx), w = (int)((i / ndx) % ndy), h = (int)(i / ((size_t)ndx * ndy));
    int8_t v = 0;
    const int ol = l + nox - oox, ow = w + noy - ooy, oh = h + noz - ooz;
    if (ol >= 0 && ow >= 0 && oh >= 0 && ol < odx && ow < ody && oh < odz) v = old[(size_t)ol + (size_t)odx * ow + (size_t)odx * ody * oh];
    nw[i] = v;
  }
}

}  // namespace mplx

struct mplx_grid {
  int device = 0;
  hipStream_t stream = nullptr;
  int8_t *map = nullptr, *inflated = nullptr;
  uint32_t *first = nullptr;              // scratch of addCloud(pts, ns): all 0xFFFFFFFF between calls
  unsigned long long *keymin = nullptr;   // all ~0 between calls
  int32_t dim[3] = {0, 0, 0}, origin[3] = {0, 0, 0};
  double origin_d[3] = {0, 0, 0};
  float res = 0;
  std::string err;
};
static std::string g_grid_create_error;
static int gfail(mplx_grid *g, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (g) g->err = buf; else g_grid_create_error = buf;
  return code;
}
#define GCHK(g, call)                                                                                  \
  do {                                                                                                 \
    hipError_t e__ = (call);                                                                           \
    if (e__ != hipSuccess) return gfail((g), MPLX_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e__)); \
  } while (0)
static mplx::GridDev grid_dev(const mplx_grid *g) {
  mplx::GridDev d;
  d.map = g->map; d.inflated = g->inflated;
  for (int i = 0; i < 3; i++) { d.dim[i] = g->dim[i]; d.origin_d[i] = g->origin_d[i]; }
  d.res = g->res;
  return d;
}
static size_t grid_cells(const mplx_grid *g) { return (size_t)g->dim[0] * g->dim[1] * g->dim[2]; }
static void grid_free_scratch(mplx_grid *g) {
  (void)hipFree(g->first); (void)hipFree(g->keymin);
  g->first = nullptr; g->keymin = nullptr;
}
extern "C" const char *mplx_grid_last_error(const mplx_grid *g) { return g ? g->err.c_str() : g_grid_create_error.c_str(); }

extern "C" int mplx_grid_allocate(mplx_grid *g, const double new_dim_d[3], const double new_ori_d[3], int *changed) {
  if (!g || !new_dim_d || !new_ori_d) return gfail(g, MPLX_ERR_ARG, "null argument");
  GCHK(g, hipSetDevice(g->device));
  int32_t nd[3], no[3];
  for (int i = 0; i < 3; i++) {
    nd[i] = (int32_t)(new_dim_d[i] / g->res);
    no[i] = (int32_t)(new_ori_d[i] / g->res);
  }
  if (nd[2] == 0 && no[2] == 0) nd[2] = 1;
  if (changed) *changed = 0;
  if (nd[0] == g->dim[0] && nd[1] == g->dim[1] && nd[2] == g->dim[2] && no[0] == g->origin[0] && no[1] == g->origin[1] && no[2] == g->origin[2]) return MPLX_OK;
  if (nd[0] <= 0 || nd[1] <= 0 || nd[2] <= 0) return gfail(g, MPLX_ERR_ARG, "empty grid");
  const size_t n = (size_t)nd[0] * nd[1] * nd[2];
  int8_t *nm = nullptr, *ni = nullptr;
  GCHK(g, hipMalloc((void **)&nm, n));
  if (hipMalloc((void **)&ni, n) != hipSuccess) { (void)hipFree(nm); return gfail(g, MPLX_ERR_HIP, "hipMalloc failed"); }
  if (g->map) {
    hipLaunchKernelGGL(mplx::grid_realloc_kernel, dim3(4096), dim3(256), 0, g->stream, g->map, g->dim[0], g->dim[1], g->dim[2], g->origin[0], g->origin[1], g->origin[2], nm,
                       nd[0], nd[1], nd[2], no[0], no[1], no[2]);
  } else {
    (void)hipMemsetAsync(nm, 0, n, g->stream);
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(ni, nm, n, hipMemcpyDeviceToDevice, g->stream);  // inflated_map_ = new_map
  if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
  if (e != hipSuccess) { (void)hipFree(nm); (void)hipFree(ni); return gfail(g, MPLX_ERR_HIP, "%s", hipGetErrorString(e)); }
  (void)hipFree(g->map); (void)hipFree(g->inflated);
  grid_free_scratch(g);
  g->map = nm; g->inflated = ni;
  for (int i = 0; i < 3; i++) { g->dim[i] = nd[i]; g->origin[i] = no[i]; g->origin_d[i] = new_ori_d[i]; }
  if (changed) *changed = 1;
  return MPLX_OK;
}
extern "C" int mplx_grid_create(int device, const double origin[3], const double dim[3], float res, mplx_grid **out) {
  if (!out || !origin || !dim || !(res > 0)) return gfail(nullptr, MPLX_ERR_ARG, "bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return gfail(nullptr, MPLX_ERR_HIP, "no HIP device");
  if (device < 0 || device >= ndev) return gfail(nullptr, MPLX_ERR_ARG, "bad device index");
  mplx_grid *g = new mplx_grid;
  g->device = device;
  g->res = res;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&g->stream) != hipSuccess) { delete g; return gfail(nullptr, MPLX_ERR_HIP, "stream creation failed"); }
  int rc = mplx_grid_allocate(g, dim, origin, nullptr);
  if (rc != MPLX_OK) { g_grid_create_error = g->err; if (g->stream) (void)hipStreamDestroy(g->stream); delete g; return rc; }
  *out = g;
  return MPLX_OK;
}
extern "C" void mplx_grid_destroy(mplx_grid *g) {
  if (!g) return;
  (void)hipSetDevice(g->device);
  (void)hipStreamSynchronize(g->stream);
  (void)hipFree(g->map); (void)hipFree(g->inflated);
  grid_free_scratch(g);
  (void)hipStreamDestroy(g->stream);
  delete g;
}
extern "C" int mplx_grid_info(const mplx_grid *g, int32_t dim[3], double origin_d[3], float *res) {
  if (!g) return MPLX_ERR_ARG;
  for (int i = 0; i < 3; i++) { if (dim) dim[i] = g->dim[i]; if (origin_d) origin_d[i] = g->origin_d[i]; }
  if (res) *res = g->res;
  return MPLX_OK;
}
extern "C" int mplx_grid_clear(mplx_grid *g) {
  if (!g || !g->map) return gfail(g, MPLX_ERR_ARG, "no grid");
  GCHK(g, hipSetDevice(g->device));
  GCHK(g, hipMemsetAsync(g->map, 0, grid_cells(g), g->stream));
  GCHK(g, hipMemsetAsync(g->inflated, 0, grid_cells(g), g->stream));
  GCHK(g, hipStreamSynchronize(g->stream));
  return MPLX_OK;
}
static int grid_upload_pts(mplx_grid *g, int n, const double *pts, double **d) {
  GCHK(g, hipMalloc((void **)d, sizeof(double) * 3 * (size_t)n));
  hipError_t e = hipMemcpyAsync(*d, pts, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, g->stream);
  if (e != hipSuccess) { (void)hipFree(*d); *d = nullptr; return gfail(g, MPLX_ERR_HIP, "%s", hipGetErrorString(e)); }
  return MPLX_OK;
}
extern "C" int mplx_grid_add_cloud(mplx_grid *g, int n, const double *pts) {
  if (!g || !g->map || n < 0 || (n > 0 && !pts)) return gfail(g, MPLX_ERR_ARG, "bad argument");
  if (n == 0) return MPLX_OK;
  GCHK(g, hipSetDevice(g->device));
  double *d = nullptr;
  int rc = grid_upload_pts(g, n, pts, &d);
  if (rc) return rc;
  hipLaunchKernelGGL(mplx::grid_add_cloud_kernel, dim3((n + 255) / 256), dim3(256), 0, g->stream, grid_dev(g), n, d);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return gfail(g, MPLX_ERR_HIP, "%s", hipGetErrorString(e));
  return MPLX_OK;
}
extern "C" int mplx_grid_add_cloud_inflate(mplx_grid *g, int n, const double *pts, int n_ns, const int32_t *ns, int32_t *new_obs, int cap, int *n_new) {
  if (!g || !g->map || n < 0 || n_ns < 0 || (n > 0 && !pts) || (n_ns > 0 && !ns) || !n_new || cap < 0 || (cap > 0 && !new_obs)) return gfail(g, MPLX_ERR_ARG, "bad argument");
  *n_new = 0;
  if (n == 0) return MPLX_OK;
  GCHK(g, hipSetDevice(g->device));
  const size_t cells = grid_cells(g);
  // per-cell scratch (first point of a cell, winning neighbour offer): both arrays exist and hold their
  // sentinels (all ones) between calls, or neither exists.  Any failure below drops both, so that the
  // next call rebuilds them instead of running the kernels on half-initialised scratch.
  auto drop_scratch = [&]() {
    (void)hipFree(g->first);
    (void)hipFree(g->keymin);
    g->first = nullptr;
    g->keymin = nullptr;
  };
  if (!g->first || !g->keymin) {
    drop_scratch();
    hipError_t es = hipMalloc((void **)&g->first, sizeof(uint32_t) * cells);
    if (es == hipSuccess) es = hipMalloc((void **)&g->keymin, sizeof(unsigned long long) * cells);
    if (es == hipSuccess) es = hipMemsetAsync(g->first, 0xFF, sizeof(uint32_t) * cells, g->stream);
    if (es == hipSuccess) es = hipMemsetAsync(g->keymin, 0xFF, sizeof(unsigned long long) * cells, g->stream);
    if (es != hipSuccess) {
      drop_scratch();
      return gfail(g, MPLX_ERR_HIP, "scratch allocation failed: %s", hipGetErrorString(es));
    }
  }
  double *d = nullptr;
  int rc = grid_upload_pts(g, n, pts, &d);
  if (rc) return rc;
  const long long pairs = (long long)n * (n_ns > 0 ? n_ns : 0);
  int32_t *dns = nullptr, *ocell = nullptr;
  unsigned long long *okey = nullptr;
  unsigned int *on = nullptr;
  hipError_t e = hipMalloc((void **)&on, sizeof(unsigned int));
  if (e == hipSuccess) e = hipMemsetAsync(on, 0, sizeof(unsigned int), g->stream);
  if (e == hipSuccess && pairs > 0) {
    e = hipMalloc((void **)&dns, sizeof(int32_t) * 3 * (size_t)n_ns);
    if (e == hipSuccess) e = hipMemcpyAsync(dns, ns, sizeof(int32_t) * 3 * (size_t)n_ns, hipMemcpyHostToDevice, g->stream);
    if (e == hipSuccess) e = hipMalloc((void **)&okey, sizeof(unsigned long long) * (size_t)pairs);
    if (e == hipSuccess) e = hipMalloc((void **)&ocell, sizeof(int32_t) * 3 * (size_t)pairs);
  }
  unsigned int n_out = 0;
  std::vector<unsigned long long> hkey;
  std::vector<int32_t> hcell;
  if (e == hipSuccess) {
    const mplx::GridDev gd = grid_dev(g);
    hipLaunchKernelGGL(mplx::grid_ns_first_kernel, dim3((n + 255) / 256), dim3(256), 0, g->stream, gd, n, d, g->first);
    if (pairs > 0) {
      const unsigned nb = (unsigned)((pairs + 255) / 256);
      hipLaunchKernelGGL(mplx::grid_ns_offer_kernel, dim3(nb), dim3(256), 0, g->stream, gd, n, d, n_ns, dns, g->first, g->keymin);
      hipLaunchKernelGGL(mplx::grid_ns_emit_kernel, dim3(nb), dim3(256), 0, g->stream, gd, n, d, n_ns, dns, g->first, g->keymin, okey, ocell, on);
    }
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&n_out, on, sizeof(unsigned int), hipMemcpyDeviceToHost, g->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    if (e == hipSuccess) {
      const long long m = (long long)n_out > (long long)n ? (long long)n_out : (long long)n;
      hipLaunchKernelGGL(mplx::grid_ns_apply_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, g->stream, gd, n, d, n_ns, dns, g->first, g->keymin, ocell, n_out);
      e = hipGetLastError();
    }
    if (e == hipSuccess && n_out > 0) {
      hkey.resize(n_out);
      hcell.resize(3 * (size_t)n_out);
      e = hipMemcpyAsync(hkey.data(), okey, sizeof(unsigned long long) * n_out, hipMemcpyDeviceToHost, g->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(hcell.data(), ocell, sizeof(int32_t) * 3 * (size_t)n_out, hipMemcpyDeviceToHost, g->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
  }
  (void)hipFree(d); (void)hipFree(dns); (void)hipFree(okey); (void)hipFree(ocell); (void)hipFree(on);
  if (e != hipSuccess) {
    (void)hipStreamSynchronize(g->stream);
    drop_scratch();  // the sentinels may not have been restored by the apply kernel
    return gfail(g, MPLX_ERR_HIP, "%s", hipGetErrorString(e));
  }
  // the sequential loop's order: by (point index, neighbour index) of the flip
  std::vector<uint32_t> ord(n_out);
  for (uint32_t i = 0; i < n_out; i++) ord[i] = i;
  std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return hkey[a] < hkey[b]; });
  for (uint32_t i = 0; i < n_out && (int)i < cap; i++)
    for (int k = 0; k < 3; k++) new_obs[3 * (size_t)i + k] = hcell[3 * (size_t)ord[i] + k];
  *n_new = (int)n_out;
  return MPLX_OK;
}
extern "C" int mplx_grid_decay(mplx_grid *g) {
  if (!g || !g->map) return gfail(g, MPLX_ERR_ARG, "no grid");
  GCHK(g, hipSetDevice(g->device));
  hipLaunchKernelGGL(mplx::grid_decay_kernel, dim3(4096), dim3(256), 0, g->stream, g->map, g->inflated, grid_cells(g));
  GCHK(g, hipGetLastError());
  GCHK(g, hipStreamSynchronize(g->stream));
  return MPLX_OK;
}
static int grid_column(mplx_grid *g, int nx, int ny, int8_t val) {
  if (!g || !g->map) return gfail(g, MPLX_ERR_ARG, "no grid");
  if (nx < 0 || nx >= g->dim[0] || ny < 0 || ny >= g->dim[1]) return MPLX_OK;  // fill(nx, ny) ignores them (voxel_grid.cpp:35-39)
  GCHK(g, hipSetDevice(g->device));
  hipLaunchKernelGGL(mplx::grid_column_kernel, dim3((g->dim[2] + 255) / 256), dim3(256), 0, g->stream, grid_dev(g), nx, ny, val);
  GCHK(g, hipGetLastError());
  GCHK(g, hipStreamSynchronize(g->stream));
  return MPLX_OK;
}
extern "C" int mplx_grid_clear_column(mplx_grid *g, int nx, int ny) { return grid_column(g, nx, ny, 0); }
extern "C" int mplx_grid_fill_column(mplx_grid *g, int nx, int ny) { return grid_column(g, nx, ny, 100); }
extern "C" int mplx_grid_fill_cell(mplx_grid *g, int nx, int ny, int nz) {
  if (!g || !g->map) return gfail(g, MPLX_ERR_ARG, "no grid");
  if (nx < 0 || nx >= g->dim[0] || ny < 0 || ny >= g->dim[1] || nz < 0 || nz >= g->dim[2]) return MPLX_OK;
  GCHK(g, hipSetDevice(g->device));
  GCHK(g, hipMemsetAsync(g->map + ((size_t)nx + (size_t)g->dim[0] * ny + (size_t)g->dim[0] * g->dim[1] * nz), 100, 1, g->stream));
  GCHK(g, hipStreamSynchronize(g->stream));
  return MPLX_OK;
}
extern "C" int mplx_grid_get_map(mplx_grid *g, int inflated, int8_t *data) {
  if (!g || !g->map || !data) return gfail(g, MPLX_ERR_ARG, "bad argument");
  GCHK(g, hipSetDevice(g->device));
  const size_t n = grid_cells(g);
  int8_t *tmp = nullptr;
  GCHK(g, hipMalloc((void **)&tmp, n));
  hipLaunchKernelGGL(mplx::grid_get_map_kernel, dim3(4096), dim3(256), 0, g->stream, inflated ? g->inflated : g->map, tmp, n);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(data, tmp, n, hipMemcpyDeviceToHost, g->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
  (void)hipFree(tmp);
  if (e != hipSuccess) return gfail(g, MPLX_ERR_HIP, "%s", hipGetErrorString(e));
  return MPLX_OK;
}
// getMap() straight into a planner context, device to device (MapUtil::setMap without PCIe)
extern "C" int mplx_grid_to_map(mplx_grid *g, int inflated, mplx_ctx *c) {
  if (!g || !g->map || !c) return gfail(g, MPLX_ERR_ARG, "bad argument");
  if (c->device != g->device) return gfail(g, MPLX_ERR_ARG, "grid and planner context live on different devices");
  GCHK(g, hipSetDevice(g->device));
  const size_t n = grid_cells(g);
  int8_t *tmp = nullptr;
  GCHK(g, hipMalloc((void **)&tmp, n));
  hipLaunchKernelGGL(mplx::grid_get_map_kernel, dim3(4096), dim3(256), 0, g->stream, inflated ? g->inflated : g->map, tmp, n);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
  if (e != hipSuccess) { (void)hipFree(tmp); return gfail(g, MPLX_ERR_HIP, "%s", hipGetErrorString(e)); }
  int rc = set_map_meta(c, g->dim, g->origin_d, (double)g->res);
  if (rc == MPLX_OK) {
    if (c->own_map) (void)hipFree(c->map);
    c->map = tmp;  // the context owns the buffer from here on
    c->own_map = true;
    rc = build_bricks(c);
  } else {
    (void)hipFree(tmp);
  }
  if (rc != MPLX_OK) return gfail(g, rc, "%s", mplx_last_error(c));
  return MPLX_OK;
}
extern "C" int mplx_grid_get_cloud(mplx_grid *g, double *pts, uint64_t cap, uint64_t *n_out) {
  if (!g || !g->map || !n_out || (cap > 0 && !pts)) return gfail(g, MPLX_ERR_ARG, "bad argument");
  GCHK(g, hipSetDevice(g->device));
  mplx::MapDev m;
  m.data = g->map; m.bricks = nullptr;
  for (int i = 0; i < 3; i++) { m.dim[i] = g->dim[i]; m.nb[i] = 0; m.origin[i] = g->origin_d[i]; }
  m.res = (double)g->res;
  const int ncol = g->dim[0] * g->dim[1];
  uint32_t *counts = nullptr;
  unsigned long long *offs = nullptr, *dtotal = nullptr;
  double *dpts = nullptr;
  hipError_t e = hipMalloc((void **)&counts, sizeof(uint32_t) * (size_t)ncol);
  if (e == hipSuccess) e = hipMalloc((void **)&offs, sizeof(unsigned long long) * (size_t)ncol);
  if (e == hipSuccess) e = hipMalloc((void **)&dtotal, sizeof(unsigned long long));
  unsigned long long total = 0;
  const int nblk = (ncol + 255) / 256 < 4096 ? (ncol + 255) / 256 : 4096;
  if (e == hipSuccess) {
    hipLaunchKernelGGL(mplx::cloud_count_kernel, dim3(nblk), dim3(256), 0, g->stream, m, 0, counts);
    hipLaunchKernelGGL(mplx::cloud_scan_kernel, dim3(1), dim3(1024), 0, g->stream, counts, offs, ncol, dtotal);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(&total, dtotal, sizeof(total), hipMemcpyDeviceToHost, g->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
  const uint64_t nw = total < cap ? total : cap;
  if (e == hipSuccess && nw > 0) {
    e = hipMalloc((void **)&dpts, sizeof(double) * 3 * nw);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(mplx::cloud_write_kernel, dim3(nblk), dim3(256), 0, g->stream, m, 0, offs, (unsigned long long)nw, dpts);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(pts, dpts, sizeof(double) * 3 * nw, hipMemcpyDeviceToHost, g->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
  }
  (void)hipFree(counts); (void)hipFree(offs); (void)hipFree(dtotal); (void)hipFree(dpts);
  if (e != hipSuccess) return gfail(g, MPLX_ERR_HIP, "%s", hipGetErrorString(e));
  *n_out = total;
  return MPLX_OK;
}
