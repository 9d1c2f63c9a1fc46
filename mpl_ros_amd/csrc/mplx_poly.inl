// mplx_poly.inl -- host side + kernels of the moving-obstacle (PolyMap) environment (C-ABI mplx_poly_*, include/mplx.h).
// Included by mplx_api.hip.  Device arithmetic: mplx_poly_dev.h.

// The kernels of this environment (poly_get_succ_kernel, astar_poly_kernel: mplx_poly_search.h) are instantiated in
// their own translation unit, mplx_poly_launch.hip, so that the device code of libmplx.so compiles in parallel.
bool mplx_launch_poly_get_succ(bool general, int grid, hipStream_t s, const mplx::PolyDev &D, int K, const int32_t *world_of, const double *states, mplx::PolySuccOut *out, int32_t *flags);
bool mplx_launch_poly_search(int control, bool general, int block, int grid, hipStream_t s, const mplx::SearchParams &P);

#include "mplx_poly_lpa_host.h"
struct mplx_poly {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  // configuration
  bool have_cfg = false;
  int control = 0, n_u = 0;
  int helpers = -1;              // look-ahead helper workgroups per leader: -1 auto (what the idle compute units allow, at most 4), 0 off, <= 15
  unsigned long long *d_help_mask = nullptr, *d_help_pub = nullptr;
  double *d_help_ring = nullptr;
  size_t help_mask_n = 0;
  int help_ring_slots = 0, last_n_help = 0;
  hipStream_t help_stream = nullptr;  // the helpers' launch (concurrent with the leaders' on the context's stream)
  hipEvent_t help_ev = nullptr;
  bool any_high_degree = false;  // an obstacle trajectory has a segment above degree two (set by mplx_poly_add_nonlinear, cleared by mplx_poly_begin)
  double dt = 1, v_max = -1, a_max = -1, j_max = -1, w = 10;
  std::vector<double> U;
  // worlds being assembled on the host
  std::vector<mplx::PolyHP> hps;
  std::vector<mplx::PolySeg> segs;
  std::vector<mplx::PolyObs> obs;       // grouped by world at commit
  std::vector<int> obs_world;
  std::vector<mplx::PolyWorld> worlds;
  bool committed = false;
  uint64_t commit_epoch = 0;  // counts mplx_poly_commit calls: what a state space's stored collision outcomes are synchronised with (mplx_plpa_*)
  // device copies
  mplx::PolyHP *d_hps = nullptr;
  mplx::PolySeg *d_segs = nullptr;
  mplx::PolyObs *d_obs = nullptr;
  mplx::PolyWorld *d_worlds = nullptr;
  double *d_U = nullptr;
  // search: pools, batch buffers and result getters of an internal planner context
  mplx_ctx *ctx = nullptr;
  int32_t *d_world_of = nullptr;
  int world_cap = 0;
  mplx::PolyPrep *d_prep_cache = nullptr;  // per workgroup x time level x obstacle (mplx_poly_dev.h)
  int prep_cache_slots = 0;
};

static int pfail(mplx_poly *p, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (p) p->err = buf; else g_create_error = buf;
  return code;
}
#define PCHK(p, call)                                                                           \
  do {                                                                                          \
    hipError_t e__ = (call);                                                                    \
    if (e__ != hipSuccess) return pfail((p), MPLX_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e__)); \
  } while (0)

extern "C" int mplx_poly_create(int device, mplx_poly **out) {
  if (!out) return pfail(nullptr, MPLX_ERR_ARG, "out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return pfail(nullptr, MPLX_ERR_HIP, "no HIP device available (%s)", hipGetErrorString(e));
  if (device < 0 || device >= n) return pfail(nullptr, MPLX_ERR_ARG, "device %d out of range", device);
  mplx_poly *p = new mplx_poly();
  p->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&p->stream) != hipSuccess || mplx_ctx_create(device, &p->ctx) != MPLX_OK) {
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
    return pfail(nullptr, MPLX_ERR_HIP, "stream / context creation failed");
  }
  *out = p;
  return MPLX_OK;
}
static void poly_free_dev(mplx_poly *p) {
  (void)hipFree(p->d_hps); (void)hipFree(p->d_segs); (void)hipFree(p->d_obs); (void)hipFree(p->d_worlds);
  p->d_hps = nullptr; p->d_segs = nullptr; p->d_obs = nullptr; p->d_worlds = nullptr;
  p->committed = false;
}
extern "C" void mplx_poly_destroy(mplx_poly *p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  (void)hipStreamSynchronize(p->stream);
  poly_free_dev(p);
  (void)hipFree(p->d_U);
  (void)hipFree(p->d_world_of);
  (void)hipFree(p->d_prep_cache);
  (void)hipFree(p->d_help_mask); (void)hipFree(p->d_help_ring); (void)hipFree(p->d_help_pub);
  if (p->help_stream) (void)hipStreamDestroy(p->help_stream);
  if (p->help_ev) (void)hipEventDestroy(p->help_ev);
  mplx_ctx_destroy(p->ctx);
  (void)hipStreamDestroy(p->stream);
  delete p;
}
extern "C" const char *mplx_poly_last_error(const mplx_poly *p) { return p ? p->err.c_str() : g_create_error.c_str(); }

extern "C" int mplx_poly_config(mplx_poly *p, int32_t control, int32_t n_u, const double *U, double dt, double v_max, double a_max, double j_max, double w) {
  if (!p || !U) return pfail(p, MPLX_ERR_ARG, "null argument");
  if (!control_ok(control)) return pfail(p, MPLX_ERR_ARG, "control kind %d is not one of VEL / ACC / JRK / SNP", control);
  if (n_u <= 0 || n_u > POLY_MAX_U) return pfail(p, MPLX_ERR_ARG, "n_u must be in [1,%d]", POLY_MAX_U);
  if (!(dt > 0)) return pfail(p, MPLX_ERR_ARG, "dt must be > 0");
  PCHK(p, hipSetDevice(p->device));
  p->control = control; p->n_u = n_u; p->dt = dt; p->v_max = v_max; p->a_max = a_max; p->j_max = j_max; p->w = w;
  p->U.assign(U, U + 2 * (size_t)n_u);
  (void)hipFree(p->d_U);
  p->d_U = nullptr;
  PCHK(p, hipMalloc((void **)&p->d_U, sizeof(double) * 2 * (size_t)n_u));
  PCHK(p, hipMemcpyAsync(p->d_U, p->U.data(), sizeof(double) * 2 * (size_t)n_u, hipMemcpyHostToDevice, p->stream));
  PCHK(p, hipStreamSynchronize(p->stream));
  p->have_cfg = true;
  return MPLX_OK;
}

extern "C" int mplx_poly_begin(mplx_poly *p, int32_t n_worlds) {
  if (!p || n_worlds <= 0) return pfail(p, MPLX_ERR_ARG, "bad argument");
  p->hps.clear(); p->segs.clear(); p->obs.clear(); p->obs_world.clear();
  p->worlds.assign((size_t)n_worlds, mplx::PolyWorld());
  p->committed = false;
  p->any_high_degree = false;
  return MPLX_OK;
}
// PolyMapUtil::setBoundingBox (poly_map_util.h:40-50) + setStartTime (:19)
extern "C" int mplx_poly_set_world(mplx_poly *p, int32_t world, const double ori[2], const double dim[2], double start_t) {
  if (!p || world < 0 || world >= (int)p->worlds.size() || !ori || !dim) return pfail(p, MPLX_ERR_ARG, "bad argument");
  mplx::PolyWorld &W = p->worlds[(size_t)world];
  W.start_t = start_t;
  W.bbox[0] = mplx::PolyHP{ori[0] + 0.0, ori[1] + dim[1] / 2, -1.0, -0.0};
  W.bbox[1] = mplx::PolyHP{ori[0] + dim[0] / 2, ori[1] + 0.0, -0.0, -1.0};
  W.bbox[2] = mplx::PolyHP{(ori[0] + dim[0]) - 0.0, (ori[1] + dim[1]) - dim[1] / 2, 1.0, 0.0};
  W.bbox[3] = mplx::PolyHP{(ori[0] + dim[0]) - dim[0] / 2, (ori[1] + dim[1]) - 0.0, 0.0, 1.0};
  return MPLX_OK;
}
// bounding radius of the polyhedron {x : n_i . (x - p_i) <= 0} around the origin of its own frame (pruning only):
// largest norm of a vertex; +inf when the normals do not positively span the plane (unbounded) or no vertex is found
static double poly_radius(int n_hp, const double *hp) {
  std::vector<double> ang;
  for (int i = 0; i < n_hp; i++) ang.push_back(atan2(hp[4 * i + 3], hp[4 * i + 2]));
  std::sort(ang.begin(), ang.end());
  double gap = n_hp ? ang.front() + 2 * M_PI - ang.back() : 2 * M_PI;
  for (size_t i = 1; i < ang.size(); i++) gap = std::max(gap, ang[i] - ang[i - 1]);
  if (n_hp < 3 || gap >= M_PI - 1e-9) return INFINITY;
  double r = -1.0, scale = 1.0;
  for (int i = 0; i < n_hp; i++) scale = std::max(scale, std::max(fabs(hp[4 * i]), fabs(hp[4 * i + 1])));
  for (int i = 0; i < n_hp; i++)
    for (int k = i + 1; k < n_hp; k++) {
      const double a1 = hp[4 * i + 2], b1 = hp[4 * i + 3], c1 = a1 * hp[4 * i] + b1 * hp[4 * i + 1];
      const double a2 = hp[4 * k + 2], b2 = hp[4 * k + 3], c2 = a2 * hp[4 * k] + b2 * hp[4 * k + 1];
      const double det = a1 * b2 - a2 * b1;
      if (fabs(det) < 1e-12) continue;
      const double x = (c1 * b2 - c2 * b1) / det, y = (a1 * c2 - a2 * c1) / det;
      bool in = true;
      for (int m = 0; m < n_hp && in; m++) in = hp[4 * m + 2] * (x - hp[4 * m]) + hp[4 * m + 3] * (y - hp[4 * m + 1]) <= 1e-9 * scale;
      if (in) r = std::max(r, sqrt(x * x + y * y));
    }
  return r < 0 ? INFINITY : r;
}
static int poly_add(mplx_poly *p, int32_t world, int kind, int n_hp, const double *hp, mplx::PolyObs &o) {
  if (!p || world < 0 || world >= (int)p->worlds.size() || n_hp <= 0 || !hp) return pfail(p, MPLX_ERR_ARG, "bad argument");
  o.radius = poly_radius(n_hp, hp);
  o.kind = kind;
  o.hp_off = (int32_t)p->hps.size();
  o.n_hp = n_hp;
  for (int i = 0; i < n_hp; i++) p->hps.push_back(mplx::PolyHP{hp[4 * i], hp[4 * i + 1], hp[4 * i + 2], hp[4 * i + 3]});
  p->obs.push_back(o);
  p->obs_world.push_back(world);
  p->committed = false;
  return MPLX_OK;
}
extern "C" int mplx_poly_add_static(mplx_poly *p, int32_t world, int32_t n_hp, const double *hp, const double pt[2]) {
  mplx::PolyObs o = mplx::PolyObs();
  if (pt) { o.p[0] = pt[0]; o.p[1] = pt[1]; }
  return poly_add(p, world, 0, n_hp, hp, o);
}
extern "C" int mplx_poly_add_linear(mplx_poly *p, int32_t world, int32_t n_hp, const double *hp, const double pt[2], const double v[2], double cov_v) {
  if (!pt || !v) return pfail(p, MPLX_ERR_ARG, "null argument");
  mplx::PolyObs o = mplx::PolyObs();
  o.p[0] = pt[0]; o.p[1] = pt[1]; o.v[0] = v[0]; o.v[1] = v[1]; o.cov_v = cov_v;
  return poly_add(p, world, 1, n_hp, hp, o);
}
extern "C" int mplx_poly_add_nonlinear(mplx_poly *p, int32_t world, int32_t n_hp, const double *hp, int32_t n_seg, const double *segs, double start_t, int32_t dis_front, int32_t dis_back) {
  if (!p || n_seg < 0 || (n_seg > 0 && !segs)) return pfail(p, MPLX_ERR_ARG, "bad argument");
  mplx::PolyObs o = mplx::PolyObs();
  o.seg_off = (int32_t)p->segs.size();
  o.n_seg = n_seg;
  double total = 0.0;
  for (int i = 0; i < n_seg; i++) {
    mplx::PolySeg s;
    for (int k = 0; k < 6; k++) { s.c[0][k] = segs[13 * i + k]; s.c[1][k] = segs[13 * i + 6 + k]; }
    for (int ax = 0; ax < 2; ax++)
      if (s.c[ax][0] != 0 || s.c[ax][1] != 0 || s.c[ax][2] != 0) p->any_high_degree = true;  // (JRK / SNP robots: the general solve())
    s.T = segs[13 * i + 12];
    total = s.T + total;  // Trajectory: taus.push_back(pr.t() + taus.back())
    p->segs.push_back(s);
  }
  o.total_t = total;
  o.start_t = start_t;
  o.fast = n_seg > 0 ? 1 : 0;  // VEL / ACC segments (checked above: +0.0 leading coefficients) with positive durations
  for (int i = 0; i < n_seg; i++) {
    const mplx::PolySeg &sg = p->segs[(size_t)o.seg_off + (size_t)i];
    if (!(sg.T > 0) || !mplx::lead_pzero(sg.c[0]) || !mplx::lead_pzero(sg.c[1])) o.fast = 0;
  }
  o.dis_front = dis_front ? 1 : 0;
  o.dis_back = dis_back ? 1 : 0;
  // representative point p_ = traj.evaluate(start_t).pos (simple_obstacle.h:126); not used by the collision tests
  return poly_add(p, world, 2, n_hp, hp, o);
}
extern "C" int mplx_poly_commit(mplx_poly *p) {
  if (!p || p->worlds.empty()) return pfail(p, MPLX_ERR_ARG, "mplx_poly_begin first");
  PCHK(p, hipSetDevice(p->device));
  // group the obstacles by world, keeping the order they were added in (static, linear, nonlinear as the caller adds them)
  std::vector<mplx::PolyObs> grouped;
  for (size_t w = 0; w < p->worlds.size(); w++) {
    p->worlds[w].obs_off = (int32_t)grouped.size();
    for (size_t i = 0; i < p->obs.size(); i++)
      if (p->obs_world[i] == (int)w) grouped.push_back(p->obs[i]);
    p->worlds[w].n_obs = (int32_t)grouped.size() - p->worlds[w].obs_off;
  }
  poly_free_dev(p);
  auto up = [&](auto **d, const auto &v) -> hipError_t {
    using T = typename std::remove_reference<decltype(v)>::type::value_type;
    const size_t bytes = sizeof(T) * std::max<size_t>(v.size(), 1);
    hipError_t e = hipMalloc((void **)d, bytes);
    if (e == hipSuccess && !v.empty()) e = hipMemcpyAsync(*d, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, p->stream);
    return e;
  };
  PCHK(p, up(&p->d_hps, p->hps));
  PCHK(p, up(&p->d_segs, p->segs));
  PCHK(p, up(&p->d_obs, grouped));
  PCHK(p, up(&p->d_worlds, p->worlds));
  PCHK(p, hipStreamSynchronize(p->stream));
  p->committed = true;
  p->commit_epoch++;
  return MPLX_OK;
}
// hyperplane equations above degree two can occur: JRK / SNP primitives, or an obstacle trajectory with such segments
static bool poly_general(const mplx_poly *p) { return p->control == CTRL_JRK || p->control == CTRL_SNP || p->any_high_degree; }
static mplx::PolyDev poly_dev(const mplx_poly *p) {
  mplx::PolyDev D{};
  D.cum = nullptr;
  D.prep_cache = nullptr;
  D.hps = p->d_hps; D.segs = p->d_segs; D.obs = p->d_obs; D.worlds = p->d_worlds;
  D.control = p->control; D.n_u = p->n_u; D.U = p->d_U;
  D.dt = p->dt; D.v_max = p->v_max; D.a_max = p->a_max; D.j_max = p->j_max; D.w = p->w;
  return D;
}
static_assert(sizeof(mplx::PolySuccOut) == sizeof(mplx_poly_succ), "PolySuccOut must mirror mplx_poly_succ");

// (internal: mplx_poly_lpa_host.h) the view the LPA* translation unit works with
extern "C" int mplx_poly_internal_view(mplx_poly *p, mplx_poly_view *out) {
  if (!p || !out) return MPLX_ERR_ARG;
  if (!p->have_cfg) return pfail(p, MPLX_ERR_ARG, "mplx_poly_config first");
  if (!p->committed) return pfail(p, MPLX_ERR_ARG, "mplx_poly_commit first");
  out->dev = poly_dev(p);
  out->general = poly_general(p) ? 1 : 0;
  out->n_worlds = (int32_t)p->worlds.size();
  out->device = p->device;
  out->stream = p->ctx->stream;
  out->guard = p->ctx->guard;
  out->deadline_s = p->ctx->deadline_s;
  out->commit_epoch = p->commit_epoch;
  return MPLX_OK;
}

extern "C" int mplx_poly_get_succ_batch(mplx_poly *p, int32_t K, const int32_t *world_of, const double *states, mplx_poly_succ *out) {
  if (!p || K <= 0 || !world_of || !states || !out) return pfail(p, MPLX_ERR_ARG, "bad argument");
  if (!p->have_cfg) return pfail(p, MPLX_ERR_ARG, "mplx_poly_config first");
  if (!p->committed) return pfail(p, MPLX_ERR_ARG, "mplx_poly_commit first");
  for (int k = 0; k < K; k++)
    if (world_of[k] < 0 || world_of[k] >= (int)p->worlds.size()) return pfail(p, MPLX_ERR_ARG, "world index out of range");
  PCHK(p, hipSetDevice(p->device));
  DevBufs bufs;
  int32_t *dw = nullptr, *dflags = nullptr;
  double *ds = nullptr;
  mplx::PolySuccOut *dout = nullptr;
  const size_t no = (size_t)K * p->n_u;
  PCHK(p, bufs.alloc(&dw, sizeof(int32_t) * K));
  PCHK(p, bufs.alloc(&ds, sizeof(double) * 9 * K));
  PCHK(p, bufs.alloc(&dout, sizeof(mplx::PolySuccOut) * no));
  PCHK(p, bufs.alloc(&dflags, sizeof(int32_t)));
  PCHK(p, hipMemcpyAsync(dw, world_of, sizeof(int32_t) * K, hipMemcpyHostToDevice, p->stream));
  PCHK(p, hipMemcpyAsync(ds, states, sizeof(double) * 9 * K, hipMemcpyHostToDevice, p->stream));
  PCHK(p, hipMemsetAsync(dflags, 0, sizeof(int32_t), p->stream));
  mplx_launch_poly_get_succ(poly_general(p), K < 4096 ? K : 4096, p->stream, poly_dev(p), K, dw, ds, dout, dflags);
  PCHK(p, hipGetLastError());
  int32_t flags = 0;
  PCHK(p, hipMemcpyAsync(out, dout, sizeof(mplx::PolySuccOut) * no, hipMemcpyDeviceToHost, p->stream));
  PCHK(p, hipMemcpyAsync(&flags, dflags, sizeof(int32_t), hipMemcpyDeviceToHost, p->stream));
  PCHK(p, hipStreamSynchronize(p->stream));
  if (flags & 1) return pfail(p, MPLX_ERR_ARG, "internal: a hyperplane equation of degree > 2 was met by the quadratic-only kernel");
  return MPLX_OK;
}

// ---- PlannerBase::plan through env_poly_map for n queries in one launch (one workgroup per query): the 16 planners of
// a decentralised tick (robot.hpp:92-133, robot_team.hpp:60-66).  Query k plans in world world_of[k] from starts[k]
// (pos2 vel2 acc2 jrk2 t) to goals[k] (same layout; goals carry position and velocity).
extern "C" int mplx_poly_set_capacity(mplx_poly *p, int32_t n_slots, uint64_t total_nodes, uint64_t total_edges, uint64_t total_open_log) {
  if (!p) return MPLX_ERR_ARG;
  return mplx_set_capacity(p->ctx, n_slots, total_nodes, total_edges, total_open_log);
}
extern "C" int mplx_poly_plan_batch(mplx_poly *p, int32_t n, const int32_t *world_of, const double *starts, const double *goals, double eps, double tol_pos,
                                    double tol_vel, int32_t max_expand, int32_t heur_ignore_dynamics, mplx_result *out) {
  if (!p || n <= 0 || !world_of || !starts || !goals || !out) return pfail(p, MPLX_ERR_ARG, "bad argument");
  if (!p->have_cfg) return pfail(p, MPLX_ERR_ARG, "mplx_poly_config first");
  if (!p->committed) return pfail(p, MPLX_ERR_ARG, "mplx_poly_commit first");
  if (p->control != CTRL_ACC && p->control != CTRL_JRK)
    return pfail(p, MPLX_ERR_ARG, "the moving-obstacle search runs ACC or JRK states (time-keyed: an SNP state's key would need 13 integers; VEL states have no caller)");
  for (int k = 0; k < n; k++)
    if (world_of[k] < 0 || world_of[k] >= (int)p->worlds.size()) return pfail(p, MPLX_ERR_ARG, "world index out of range");
  mplx_ctx *c = p->ctx;
  PCHK(p, hipSetDevice(p->device));
  // the internal context carries the record layout and the pools; its own voxel set-up stays unused
  c->cfg = mplx_config();
  c->yaw = false;
  c->cfg.control = p->control;
  c->cfg.n_u = p->n_u;
  c->cfg.dt = p->dt; c->cfg.v_max = p->v_max; c->cfg.a_max = p->a_max; c->cfg.j_max = p->j_max;
  c->cfg.w = p->w; c->cfg.eps = eps;
  c->cfg.tol_pos = tol_pos; c->cfg.tol_vel = tol_vel; c->cfg.tol_acc = -1.0;
  c->cfg.t_max = INFINITY;
  c->cfg.max_expand = max_expand;
  c->cfg.heur_ignore_dynamics = heur_ignore_dynamics;
  c->U.assign(3 * (size_t)p->n_u, 0.0);
  for (int i = 0; i < p->n_u; i++) { c->U[3 * i] = p->U[2 * i]; c->U[3 * i + 1] = p->U[2 * i + 1]; }
  c->cfg.U = c->U.data();
  c->have_cfg = true;
  const int helpers_saved = c->helpers;
  c->helpers = 0;  // (no look-ahead cache arrays for this kernel)
  const int slots = n < c->n_slots ? n : c->n_slots;
  int r = ensure_pools(c, slots);
  c->helpers = helpers_saved;
  if (r != MPLX_OK) return pfail(p, r, "%s", c->err.c_str());
  if ((r = ensure_batch(c, n)) != MPLX_OK) return pfail(p, r, "%s", c->err.c_str());
  if (p->world_cap < n) {
    (void)hipFree(p->d_world_of);
    p->d_world_of = nullptr;
    PCHK(p, hipMalloc((void **)&p->d_world_of, sizeof(int32_t) * (size_t)n));
    p->world_cap = n;
  }
  std::vector<QueryIn> in((size_t)n);
  std::vector<int32_t> order((size_t)n);
  for (int k = 0; k < n; k++) {
    const double *s = starts + 9 * (size_t)k, *g = goals + 9 * (size_t)k;
    QueryIn &q = in[(size_t)k];
    memset(&q, 0, sizeof(q));
    q.start.p[0] = s[0]; q.start.p[1] = s[1]; q.start.v[0] = s[2]; q.start.v[1] = s[3];
    q.goal.p[0] = g[0]; q.goal.p[1] = g[1]; q.goal.v[0] = g[2]; q.goal.v[1] = g[3];
    if (p->control == CTRL_JRK) { q.start.a[0] = s[4]; q.start.a[1] = s[5]; q.goal.a[0] = g[4]; q.goal.a[1] = g[5]; }
    q.start_t = s[8];
    q.goal_control = p->control;
    order[(size_t)k] = k;
  }
  SearchParams P = c->pools;
  fill_params(c, P);
  P.boxes = nullptr;
  P.cap_rec = c->cap_rec;
  P.nq = n;
  P.queries = c->d_in;
  P.order = c->d_order;
  P.out = c->d_out;
  P.traj_nodes = c->d_traj_nodes; P.traj_actions = c->d_traj_actions; P.traj_states = c->d_traj_states;
  P.rec_ids = c->cap_rec ? c->d_rec : nullptr;
  P.node_tables = c->d_node_tables;
  P.edge_tables = c->d_edge_tables;
  P.next_query = c->d_next;
  P.poly = poly_dev(p);
  P.poly_world = p->d_world_of;
  // look-ahead helpers: only when every leader has exactly one query (the batched tick) and the masks fit one word
  int n_help = 0;
  if (p->helpers != 0 && n <= slots && p->n_u <= 31 && c->n_cus > slots) {
    n_help = (c->n_cus - slots) / slots;
    const int want = p->helpers < 0 ? 4 : p->helpers;  // (measured on the Team2 tick: 3 per robot already serve 99.9 % of the pops)
    if (n_help > want) n_help = want;
  }
  const int grid = slots * (1 + n_help);
  constexpr int RING_LOG = 16;
  if (n_help > 0) {
    const size_t recs = (size_t)P.node_chunks << NODE_CH_LOG;
    if (p->help_mask_n < recs) {
      (void)hipFree(p->d_help_mask);
      p->d_help_mask = nullptr;
      PCHK(p, hipMalloc((void **)&p->d_help_mask, sizeof(unsigned long long) * recs));
      p->help_mask_n = recs;
    }
    if (p->help_ring_slots < slots) {
      (void)hipFree(p->d_help_ring); (void)hipFree(p->d_help_pub);
      p->d_help_ring = nullptr; p->d_help_pub = nullptr;
      PCHK(p, hipMalloc((void **)&p->d_help_ring, sizeof(double) * 8 * ((size_t)slots << RING_LOG)));
      PCHK(p, hipMalloc((void **)&p->d_help_pub, sizeof(unsigned long long) * (size_t)slots));
      p->help_ring_slots = slots;
    }
    PCHK(p, hipMemsetAsync(p->d_help_mask, 0, sizeof(unsigned long long) * recs, c->stream));
    PCHK(p, hipMemsetAsync(p->d_help_pub, 0, sizeof(unsigned long long) * (size_t)slots, c->stream));
    // (the ring needs no clearing: a helper reads entry `id` only after the leader's count has passed it, and the entry
    //  names its state)
  }
  P.poly.help_mask = p->d_help_mask; P.poly.help_ring = p->d_help_ring; P.poly.help_pub = p->d_help_pub;
  P.poly.help_ring_log = RING_LOG; P.poly.n_help = n_help;
  P.help_lead = slots;
  p->last_n_help = n_help;
  if (n_help > 0 && !p->help_stream) {
    PCHK(p, hipStreamCreateWithFlags(&p->help_stream, hipStreamNonBlocking));
    PCHK(p, hipEventCreateWithFlags(&p->help_ev, hipEventDisableTiming));
  }
  {
    const size_t per = (size_t)mplx::POLY_CACHE_LEVELS * mplx::POLY_MAX_OBS;
    if (p->prep_cache_slots < grid) {  // (one slice per workgroup of the launch, helpers included)
      (void)hipFree(p->d_prep_cache);
      p->d_prep_cache = nullptr;
      PCHK(p, hipMalloc((void **)&p->d_prep_cache, sizeof(mplx::PolyPrep) * per * (size_t)grid));
      p->prep_cache_slots = grid;
    }
    PCHK(p, hipMemsetAsync(p->d_prep_cache, 0, sizeof(mplx::PolyPrep) * per * (size_t)grid, c->stream));  // tag_q 0: empty
    P.poly.prep_cache = p->d_prep_cache;
  }
  hipStream_t st = c->stream;
  PCHK(p, hipMemcpyAsync(p->d_world_of, world_of, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, st));
  PCHK(p, hipMemcpyAsync(c->d_order, order.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, st));
  if (int rt = table_prepare(c, P, st)) return pfail(p, rt, "%s", c->err.c_str());
  PCHK(p, hipMemsetAsync(P.chunk_next, 0, 4 * sizeof(uint32_t), st));
  PCHK(p, hipMemcpyAsync(c->d_in, in.data(), sizeof(QueryIn) * (size_t)n, hipMemcpyHostToDevice, st));
  PCHK(p, hipMemsetAsync(c->d_next, 0, sizeof(int32_t), st));
  guard_arm(c);
  PCHK(p, hipEventRecord(c->ev0, st));
  if (n_help > 0) {
    // leaders: one 64-lane workgroup per query on the context's stream; helpers: 256-lane workgroups in a launch of
    // their own, concurrent on a second stream (it starts behind the memsets above; a leader never waits for it)
    PCHK(p, hipEventRecord(p->help_ev, st));
    PCHK(p, hipStreamWaitEvent(p->help_stream, p->help_ev, 0));
    SearchParams PH = P;
    PH.help_lead = 0;
    PH.poly.prep_slice0 = slots;  // (the leaders' launch owns slices 0 .. slots - 1)
    mplx_launch_poly_search(p->control, poly_general(p), 256, slots * n_help, p->help_stream, PH);
    PCHK(p, hipEventRecord(c->ev0, st));
    mplx_launch_poly_search(p->control, poly_general(p), 64, slots, st, P);
  } else {
    mplx_launch_poly_search(p->control, poly_general(p), 256, slots, st, P);
  }
  (void)grid;
  PCHK(p, hipGetLastError());
  PCHK(p, hipEventRecord(c->ev1, st));
  c->last_out.resize((size_t)n);
  // (the guarded waits first: a device-to-host copy into pageable memory would block the host until the stream has drained)
  // (one deadline for both waits -- guard_wait counts from the launch -- and an aborted leader launch does not return before
  //  the helper launch has been drained too, with the abort word still raised: a helper that has not polled it yet must see it)
  if (int rw = guard_wait(c, st, "the moving-obstacle search launch", n_help > 0)) {
    if (n_help > 0 && !c->wedged) {
      const std::string first = c->err;
      (void)guard_wait(c, p->help_stream, "the moving-obstacle helper launch");  // (bounded by the same deadline + grace; lowers the abort word)
      if (!c->wedged) c->err = first;
      guard_disarm(c);
    }
    c->last_nq = 0;
    return pfail(p, rw, "%s", c->err.c_str());
  }
  if (n_help > 0) {  // (the helpers leave when their leader has published DONE)
    if (int rw = guard_wait(c, p->help_stream, "the moving-obstacle helper launch")) {
      c->last_nq = 0;
      return pfail(p, rw, "%s", c->err.c_str());
    }
  }
  PCHK(p, hipMemcpyAsync(c->last_out.data(), c->d_out, sizeof(QueryOut) * (size_t)n, hipMemcpyDeviceToHost, st));
  PCHK(p, hipStreamSynchronize(st));
  PCHK(p, hipEventElapsedTime(&c->last_ms, c->ev0, c->ev1));
  for (int k = 0; k < n; k++) fill_result(c->last_out[(size_t)k], out[k]);
  c->last_nq = n;
  c->last_single = (n == 1);
  c->last_control = p->control;
  c->last_yaw = false;
  c->last_dt = p->dt;
  c->last_U = c->U;
  c->plan_epoch++;
  for (int k = 0; k < n; k++)
    if (out[k].status == MPLX_PLAN_INTERNAL) return pfail(p, MPLX_ERR_ARG, "internal: a hyperplane equation of degree > 2 was met by the quadratic-only kernel");
  return MPLX_OK;
}
// trajectory of query q of the last mplx_poly_plan_batch: actions[traj_len], node_ids[traj_len + 1], states (traj_len + 1) x 9
extern "C" int mplx_poly_result_traj(mplx_poly *p, int32_t q, int32_t *actions, int32_t *node_ids, double *states) {
  if (!p) return MPLX_ERR_ARG;
  mplx_ctx *c = p->ctx;
  if (q < 0 || q >= c->last_nq) return pfail(p, MPLX_ERR_ARG, "no such query");
  const int len = c->last_out[(size_t)q].traj_len;
  if (c->last_out[(size_t)q].status != MPLX_PLAN_OK || len <= 0) return MPLX_OK;
  std::vector<mplx_waypoint> wps((size_t)len + 1);
  int r = mplx_result_traj(c, q, nullptr, wps.data(), actions, node_ids);
  if (r) return pfail(p, r, "%s", c->err.c_str());
  if (states)
    for (int i = 0; i <= len; i++) {
      double *s = states + 9 * (size_t)i;
      s[0] = wps[(size_t)i].pos[0]; s[1] = wps[(size_t)i].pos[1]; s[2] = wps[(size_t)i].vel[0]; s[3] = wps[(size_t)i].vel[1];
      s[4] = wps[(size_t)i].acc[0]; s[5] = wps[(size_t)i].acc[1];
      s[6] = s[7] = 0.0;
      s[8] = wps[(size_t)i].t;
    }
  return MPLX_OK;
}
// shader-clock cycles query q of the last batch spent in [0] pop, [1] get_succ (primitives + collide), [2] look-up + commit
extern "C" int mplx_poly_result_cycles(mplx_poly *p, int32_t q, uint64_t cyc[10]) { return p ? mplx_result_cycles(p->ctx, q, cyc) : MPLX_ERR_ARG; }
extern "C" int mplx_poly_last_kernel_ms(const mplx_poly *p, float *ms) { return p ? mplx_last_kernel_ms(p->ctx, ms) : MPLX_ERR_ARG; }
extern "C" int mplx_poly_result_expanded(mplx_poly *p, int32_t q, uint32_t cap, int32_t *ids, uint32_t *n) {
  if (!p) return MPLX_ERR_ARG;
  int r = mplx_result_expanded(p->ctx, q, cap, ids, n);
  return r ? pfail(p, r, "%s", p->ctx->err.c_str()) : MPLX_OK;
}
extern "C" int mplx_poly_set_record(mplx_poly *p, uint32_t cap) { return p ? mplx_set_record(p->ctx, cap) : MPLX_ERR_ARG; }
extern "C" int mplx_poly_set_helpers(mplx_poly *p, int32_t per_leader) {
  if (!p || per_leader < -1 || per_leader > 15) return pfail(p, MPLX_ERR_ARG, "helpers per leader: -1 (auto), 0 (off) .. 15");
  p->helpers = per_leader;
  return MPLX_OK;
}
extern "C" int mplx_poly_last_helpers(const mplx_poly *p) { return p ? p->last_n_help : 0; }
// launch guard of the moving-obstacle search (mplx_set_deadline of the planner's internal context)
extern "C" int mplx_poly_set_deadline(mplx_poly *p, double seconds) { return p ? mplx_set_deadline(p->ctx, seconds) : MPLX_ERR_ARG; }
