// mplx_lpa.inl -- host side of the LPA* planner objects (mplx_lpa_* of include/mplx.h); included by mplx_api.hip.
// An mplx_lpa is the state space of ONE incremental planner on a context's map: its own flat pools (two sets, for
// getSubStateSpace), so that an A* planner sharing the MapUtil / context (planner_ and replan_planner_ of
// map_replanner_node.cpp:415,427) never touches it.  Kernels: mplx_lpa.h.

namespace mplx { struct LpaParams; }
bool mplx_launch_lpa(int what, int mode, hipStream_t s, const mplx::SearchParams &P, const mplx::LpaParams &A, int pass = 0, int grid = 1);
#include "mplx_lpa.h"
bool mplx_launch_lpa_import(hipStream_t s, const mplx::SearchParams &P, const mplx::LpaParams &A, const mplx::LpaImportArgs &I, int wide);

struct LpaSpace {
  char *node_pool = nullptr, *edge_pool = nullptr;
  unsigned long long *table = nullptr;
  uint2 *blocked_log = nullptr;
  LpaState *st = nullptr;
};

struct mplx_lpa {
  mplx_ctx *ctx = nullptr;
  std::string err;
  uint64_t cap_nodes = 1u << 20, cap_edges = 1u << 22, cap_log = 1u << 21;
  uint32_t cap_rec = 0;
  // pools
  bool pools_valid = false;
  int pool_control = 0;
  uint64_t pool_nodes = 0, pool_edges = 0, pool_log = 0, table_slots = 0;
  LpaSpace sp[2];
  int cur = 0;
  char *open_pool = nullptr;
  uint32_t *bkt_head = nullptr;
  QueryIn *d_in = nullptr;
  QueryOut *d_out = nullptr;
  int32_t *d_traj_nodes = nullptr, *d_traj_actions = nullptr, *d_rec = nullptr;
  double *d_traj_states = nullptr;
  uint32_t rec_alloc = 0;
  // host view
  bool valid = false;            // a state space exists (PlannerBase::initialized())
  mplx_config cfg{};             // the set-up the state space was built with
  std::vector<double> U;
  mplx_waypoint goal{};
  int32_t root_key[MAX_KEY] = {};
  QueryOut last_out{};
  int reroot_mode = 2;  // getSubStateSpace: 0 Dijkstra through the expanded states, 1 plan afresh from the new root, 2 auto (mplx_lpa_set_reroot)
  // what the import lane was last configured for (lpa_import_lane): the parent's cfg_epoch, 1 plan / 2 Dijkstra, the handle's capacities
  uint64_t imp_cfg_epoch = ~0ull;
  int imp_mode = 0;
  uint64_t imp_cap_nodes = 0, imp_cap_edges = 0, imp_cap_log = 0;
  LpaState st{};                 // host copy of the current space's scalars
  // the last trajectory (getTraj() returns the Trajectory stored at plan time, also after getSubStateSpace)
  int traj_len = 0;
  std::vector<int32_t> traj_nodes, traj_actions;
  std::vector<double> traj_states;
  float last_ms = 0;
  // a FRESH plan is an A* from scratch: planned by the speculative kernel on a private lane context (the parent's map replica and
  // planner set-up, like a lane of an mplx_stream) and imported into the LPA* pools (mplx_lpa.h "import of a finished A*")
  mplx_ctx *imp = nullptr;
  uint64_t imp_map_epoch = 0;
  uint32_t *d_blk_off = nullptr;
  unsigned long long *d_blk_mask = nullptr;
  size_t blk_cap = 0;
  hipEvent_t imp_ev0 = nullptr, imp_ev1 = nullptr;
};

static int lfail(mplx_lpa *l, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (l) l->err = buf;
  return code;
}
#define LCHK(l, call)                                                                                        \
  do {                                                                                                       \
    hipError_t e__ = (call);                                                                                 \
    if (e__ != hipSuccess) return lfail((l), MPLX_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e__)); \
  } while (0)

static void lpa_free_pools(mplx_lpa *l) {
  for (int k = 0; k < 2; k++) {
    (void)hipFree(l->sp[k].node_pool); (void)hipFree(l->sp[k].edge_pool); (void)hipFree(l->sp[k].table);
    (void)hipFree(l->sp[k].blocked_log); (void)hipFree(l->sp[k].st);
    l->sp[k] = LpaSpace();
  }
  (void)hipFree(l->open_pool); (void)hipFree(l->bkt_head);
  l->open_pool = nullptr; l->bkt_head = nullptr;
  l->pools_valid = false;
  l->valid = false;
}

extern "C" int mplx_lpa_create(mplx_ctx *ctx, mplx_lpa **out) {
  if (!ctx || !out) return MPLX_ERR_ARG;
  *out = new mplx_lpa();
  (*out)->ctx = ctx;
  return MPLX_OK;
}
extern "C" void mplx_lpa_destroy(mplx_lpa *l) {
  if (!l) return;
  (void)hipSetDevice(l->ctx->device);
  (void)hipStreamSynchronize(l->ctx->stream);
  lpa_free_pools(l);
  (void)hipFree(l->d_in); (void)hipFree(l->d_out); (void)hipFree(l->d_traj_nodes); (void)hipFree(l->d_traj_actions);
  (void)hipFree(l->d_traj_states); (void)hipFree(l->d_rec);
  (void)hipFree(l->d_blk_off); (void)hipFree(l->d_blk_mask);
  if (l->imp_ev0) (void)hipEventDestroy(l->imp_ev0);
  if (l->imp_ev1) (void)hipEventDestroy(l->imp_ev1);
  if (l->imp) mplx_ctx_destroy(l->imp);
  delete l;
}
extern "C" const char *mplx_lpa_last_error(const mplx_lpa *l) { return l ? l->err.c_str() : ""; }
extern "C" int mplx_lpa_set_capacity(mplx_lpa *l, uint64_t nodes, uint64_t edges, uint64_t open_log) {
  if (!l) return MPLX_ERR_ARG;
  if (nodes) l->cap_nodes = nodes;
  if (edges) l->cap_edges = edges;
  if (open_log) l->cap_log = open_log;
  return MPLX_OK;
}
extern "C" int mplx_lpa_set_record(mplx_lpa *l, uint32_t cap) {
  if (!l) return MPLX_ERR_ARG;
  l->cap_rec = cap;
  return MPLX_OK;
}
extern "C" int mplx_lpa_initialized(const mplx_lpa *l) { return l && l->valid ? 1 : 0; }
extern "C" int mplx_lpa_reset(mplx_lpa *l) {
  if (!l) return MPLX_ERR_ARG;
  l->valid = false;
  l->traj_len = 0;
  return MPLX_OK;
}

static int lpa_ensure(mplx_lpa *l) {
  mplx_ctx *c = l->ctx;
  const int control = c->cfg.control;
  if (l->pools_valid && l->pool_control == control && l->pool_nodes == l->cap_nodes && l->pool_edges == l->cap_edges && l->pool_log == l->cap_log) {
    if (l->cap_rec > l->rec_alloc) {
      (void)hipFree(l->d_rec);
      l->d_rec = nullptr;
      LCHK(l, hipMalloc((void **)&l->d_rec, sizeof(int32_t) * (size_t)l->cap_rec));
      l->rec_alloc = l->cap_rec;
    }
    return MPLX_OK;
  }
  lpa_free_pools(l);
  uint64_t nch = std::max<uint64_t>(1, (l->cap_nodes + (1u << NODE_CH_LOG) - 1) >> NODE_CH_LOG);
  uint64_t ech = std::max<uint64_t>(1, (l->cap_edges + (1u << EDGE_CH_LOG) - 1) >> EDGE_CH_LOG);
  uint64_t och = std::max<uint64_t>(1, (l->cap_log + (1u << OPEN_CH_LOG) - 1) >> OPEN_CH_LOG);
  if (nch > (uint64_t)MAX_NODE_CH || ech > (uint64_t)MAX_EDGE_CH || och > (uint64_t)MAX_OPEN_CH)
    return lfail(l, MPLX_ERR_ARG, "LPA* capacity too large (at most %d node / %d predecessor / %d OPEN-log chunks)", MAX_NODE_CH, MAX_EDGE_CH, MAX_OPEN_CH);
  l->table_slots = next_pow2(4ull * (nch << NODE_CH_LOG));
  for (int k = 0; k < 2; k++) {
    LCHK(l, hipMalloc((void **)&l->sp[k].node_pool, (size_t)(nch << NODE_CH_LOG) * rec_bytes(control)));
    LCHK(l, hipMalloc((void **)&l->sp[k].edge_pool, (size_t)(ech << EDGE_CH_LOG) * EDGE_BYTES));
    LCHK(l, hipMalloc((void **)&l->sp[k].table, (size_t)l->table_slots * sizeof(unsigned long long)));
    LCHK(l, hipMalloc((void **)&l->sp[k].blocked_log, (size_t)(ech << EDGE_CH_LOG) * sizeof(uint2)));
    LCHK(l, hipMalloc((void **)&l->sp[k].st, sizeof(LpaState)));
  }
  LCHK(l, hipMalloc((void **)&l->open_pool, (size_t)(och << OPEN_CH_LOG) * OPEN_BYTES));
  LCHK(l, hipMalloc((void **)&l->bkt_head, sizeof(uint32_t) * 2 * NB * NSUB));
  LCHK(l, hipMemsetAsync(l->bkt_head, 0xFF, sizeof(uint32_t) * 2 * NB * NSUB, c->stream));
  if (!l->d_in) {
    LCHK(l, hipMalloc((void **)&l->d_in, sizeof(QueryIn)));
    LCHK(l, hipMalloc((void **)&l->d_out, sizeof(QueryOut)));
    LCHK(l, hipMalloc((void **)&l->d_traj_nodes, sizeof(int32_t) * (MAX_TRAJ + 1)));
    LCHK(l, hipMalloc((void **)&l->d_traj_actions, sizeof(int32_t) * MAX_TRAJ));
    LCHK(l, hipMalloc((void **)&l->d_traj_states, sizeof(double) * (MAX_TRAJ + 1) * 13));
  }
  if (l->cap_rec > l->rec_alloc) {
    (void)hipFree(l->d_rec);
    l->d_rec = nullptr;
    LCHK(l, hipMalloc((void **)&l->d_rec, sizeof(int32_t) * (size_t)l->cap_rec));
    l->rec_alloc = l->cap_rec;
  }
  l->pool_control = control;
  l->pool_nodes = l->cap_nodes; l->pool_edges = l->cap_edges; l->pool_log = l->cap_log;
  l->pools_valid = true;
  return MPLX_OK;
}

static void lpa_params(const mplx_lpa *l, int space, SearchParams &P, LpaParams &A) {
  const mplx_ctx *c = l->ctx;
  P = SearchParams{};
  fill_params(c, P);
  // The one-workgroup kernels pop ONE state per iteration: with the batch kernels' bucket width (8 edge costs; a fine bucket is 1 / 1024
  // of it) almost every pop is a refill from the far lists.  64 edge costs measured best at C2 size (profiles/r06p_*: repair 8.25 ->
  // 6.74 ms; 512: 7.04, 4096: 10.6).  The pop order does not depend on it.  (measurement: MPLX_LPA_BUCKET_FACTOR)
  static const double bucket_factor = [] { const char *e = getenv("MPLX_LPA_BUCKET_FACTOR"); const double v = e ? atof(e) : 0.0; return v > 0 ? v : 64.0; }();
  if (c->bucket_width <= 0 && P.w * P.dt > 0) P.bucket_width = P.w * P.dt * bucket_factor;
  const LpaSpace &s = l->sp[space];
  P.node_pool = s.node_pool; P.edge_pool = s.edge_pool; P.open_pool = l->open_pool;
  P.node_chunks = (uint32_t)std::max<uint64_t>(1, (l->cap_nodes + (1u << NODE_CH_LOG) - 1) >> NODE_CH_LOG);
  P.edge_chunks = (uint32_t)std::max<uint64_t>(1, (l->cap_edges + (1u << EDGE_CH_LOG) - 1) >> EDGE_CH_LOG);
  P.open_chunks = (uint32_t)std::max<uint64_t>(1, (l->cap_log + (1u << OPEN_CH_LOG) - 1) >> OPEN_CH_LOG);
  P.table = s.table;
  P.table_mask = l->table_slots - 1;
  P.bkt_head = l->bkt_head;
  P.cap_rec = l->cap_rec;
  P.nq = 1;
  P.queries = l->d_in;
  P.out = l->d_out;
  P.traj_nodes = l->d_traj_nodes; P.traj_actions = l->d_traj_actions; P.traj_states = l->d_traj_states;
  P.rec_ids = l->cap_rec ? l->d_rec : nullptr;
  A = LpaParams{};
  A.st = s.st;
  A.blocked_log = s.blocked_log;
  A.blocked_cap = (uint32_t)std::min<uint64_t>((uint64_t)P.edge_chunks << EDGE_CH_LOG, 0xFFFFFFF0ull);
}

static bool lpa_same_setup(const mplx_lpa *l) {
  const mplx_config &a = l->cfg, &b = l->ctx->cfg;
  return a.control == b.control && a.n_u == b.n_u && a.dt == b.dt && a.v_max == b.v_max && a.a_max == b.a_max && a.j_max == b.j_max && a.w == b.w &&
         a.eps == b.eps && a.heur_ignore_dynamics == b.heur_ignore_dynamics && l->U == l->ctx->U;
}
static bool lpa_same_goal(const mplx_waypoint &a, const mplx_waypoint &b) {
  return a.control == b.control && memcmp(a.pos, b.pos, sizeof(a.pos)) == 0 && memcmp(a.vel, b.vel, sizeof(a.vel)) == 0 &&
         memcmp(a.acc, b.acc, sizeof(a.acc)) == 0 && memcmp(a.jrk, b.jrk, sizeof(a.jrk)) == 0;
}

// A fresh LPA* plan = an A* from scratch: planned by the speculative kernel (helper workgroups and all) on the handle's private
// lane context, then imported into the LPA* pools of space l->cur (P / A: lpa_params of that space).  The lane adopts the
// parent's map replica and planner set-up like a lane of an mplx_stream; it owns pools of the LPA* handle's capacity.
// the lane of an import: the parent's map replica and planner set-up (dijkstra: eps = 0, no goal, no cap -- getSubStateSpace),
// pools of the LPA* handle's capacity, an expansion record that holds the whole order
static int lpa_import_lane(mplx_lpa *l, bool dijkstra) {
  mplx_ctx *c = l->ctx;
  int r;
  if (!l->imp) {
    if ((r = mplx_ctx_create(c->device, &l->imp)) != MPLX_OK) return lfail(l, r, "LPA* import lane: %s", g_create_error.c_str());
    l->imp_map_epoch = 0;
    LCHK(l, hipEventCreate(&l->imp_ev0));
    LCHK(l, hipEventCreate(&l->imp_ev1));
  }
  mplx_ctx *lane = l->imp;
  if (l->imp_map_epoch != c->map_epoch || lane->map != c->map) {  // the parent's map changed (or was replaced): adopt it again
    if ((r = mplx_map_set_device(lane, c->map, c->dim, c->origin, c->res)) != MPLX_OK) return lfail(l, r, "LPA* import lane: %s", lane->err.c_str());
    l->imp_map_epoch = c->map_epoch;
  }
  //  the lane is re-configured -- device buffers of the lattice freed and allocated, a stream synchronisation -- only when the
  // parent's set-up changed since, or when the lane last ran in the other mode (plan / Dijkstra): not on every replanning call
  const int want_mode = dijkstra ? 2 : 1;
  if (l->imp_cfg_epoch == c->cfg_epoch && l->imp_mode == want_mode && l->imp_cap_nodes == l->cap_nodes && l->imp_cap_edges == l->cap_edges && l->imp_cap_log == l->cap_log) {
    lane->deadline_s = c->deadline_s;
    lane->helpers = c->helpers; lane->help_reserved = c->help_reserved; lane->help_rows = c->help_rows; lane->help_limit = -1;
    return MPLX_OK;
  }
  if ((r = stream_lane_setup(c, lane, false)) != MPLX_OK) return lfail(l, r, "LPA* import lane: %s", lane->err.c_str());
  if (dijkstra) {  // the same lattice and limits; key = g, run until OPEN is empty
    mplx_config cfg = lane->cfg;
    cfg.control = lane->cfg.control;
    const std::vector<double> u_copy = lane->U;  // (mplx_planner_config assigns lane->U from cfg.U: never from the vector's own storage)
    cfg.U = u_copy.data();
    cfg.U_yaw = nullptr;
    cfg.eps = 0.0;
    cfg.tol_pos = -1.0; cfg.tol_vel = -1.0; cfg.tol_acc = -1.0;  // (|dp| <= -1 never holds: no goal test ends the search)
    cfg.t_max = INFINITY;
    cfg.max_expand = -1;
    if ((r = mplx_planner_config(lane, &cfg)) != MPLX_OK) return lfail(l, r, "LPA* import lane: %s", lane->err.c_str());
  }
  lane->helpers = c->helpers; lane->help_reserved = c->help_reserved; lane->help_rows = c->help_rows; lane->help_limit = -1;
  mplx_set_capacity(lane, 1, l->cap_nodes, l->cap_edges, l->cap_log);
  lane->cap_rec = (uint32_t)std::min<uint64_t>(l->cap_nodes, 0xFFFFFFF0ull);  // every expansion closes a state: the record holds the whole order
  l->imp_cfg_epoch = c->cfg_epoch;
  l->imp_mode = want_mode;
  l->imp_cap_nodes = l->cap_nodes; l->imp_cap_edges = l->cap_edges; l->imp_cap_log = l->cap_log;
  return MPLX_OK;
}
// the three import kernels on the lane's stream (P / A: lpa_params of the space written), timed into *ms
static int lpa_import_launch(mplx_lpa *l, const SearchParams &P, const LpaParams &A, int mode, float *ms) {
  mplx_ctx *c = l->ctx, *lane = l->imp;
  const size_t n_exp = lane->last_out[0].n_recorded;
  if (l->blk_cap < n_exp + 1) {
    (void)hipFree(l->d_blk_off); (void)hipFree(l->d_blk_mask);
    l->d_blk_off = nullptr; l->d_blk_mask = nullptr;
    l->blk_cap = std::max<size_t>(n_exp + 1, (size_t)1 << 16) * 2;
    LCHK(l, hipMalloc((void **)&l->d_blk_off, sizeof(uint32_t) * l->blk_cap));
    LCHK(l, hipMalloc((void **)&l->d_blk_mask, sizeof(unsigned long long) * 2 * l->blk_cap));
  }
  LpaImportArgs I{};
  I.node_pool = lane->pools.node_pool; I.edge_pool = lane->pools.edge_pool;
  I.node_table = lane->d_node_tables; I.edge_table = lane->d_edge_tables;
  I.out = lane->d_out;
  I.rec_ids = lane->d_rec;
  I.traj_nodes = lane->d_traj_nodes; I.traj_actions = lane->d_traj_actions; I.traj_states = lane->d_traj_states;
  I.blk_off = l->d_blk_off; I.blk_mask = l->d_blk_mask;
  I.dst_rec = (mode == 0 && l->cap_rec) ? l->d_rec : nullptr; I.dst_cap_rec = mode == 0 ? l->cap_rec : 0;
  I.mode = mode;
  if (mode == 1) {  // the heuristic of a state the old space did not hold: the planner's goal
    I.hp.w = P.w; I.hp.v_max = P.v_max; I.hp.heur_ignore_dynamics = P.heur_ignore_dynamics;
    I.hp.goal_control = l->goal.control;
    wp_to_state(l->goal, I.hp.goal);
    I.hp.goal_nkey = state_key(l->goal.control, I.hp.goal, I.hp.goal_key);
  }
  hipStream_t s = lane->stream;
  LCHK(l, hipEventRecord(l->imp_ev0, s));
  LCHK(l, hipMemsetAsync(P.table, 0xFF, (size_t)l->table_slots * sizeof(unsigned long long), s));
  LCHK(l, hipMemsetAsync(A.st, 0, sizeof(LpaState), s));
  if (!mplx_launch_lpa_import(s, P, A, I, std::max(1, 4 * c->n_cus))) return lfail(l, MPLX_ERR_ARG, "LPA* supports lattices of at most 128 control inputs (got %d)", P.n_u);
  LCHK(l, hipGetLastError());
  LCHK(l, hipEventRecord(l->imp_ev1, s));
  LCHK(l, hipStreamSynchronize(s));  // (short, bounded kernels: copy, one get_succ per expanded state, a scan)
  LCHK(l, hipEventElapsedTime(ms, l->imp_ev0, l->imp_ev1));
  return MPLX_OK;
}
static int lpa_plan_by_import(mplx_lpa *l, const mplx_waypoint *start, const mplx_waypoint *goal, const SearchParams &P, const LpaParams &A) {
  int r;
  if ((r = lpa_import_lane(l, false)) != MPLX_OK) return r;
  mplx_ctx *lane = l->imp;
  mplx_result res;
  if ((r = mplx_plan(lane, start, goal, &res)) != MPLX_OK) return lfail(l, r, "%s", lane->err.c_str());
  float imp_ms = 0;
  if ((r = lpa_import_launch(l, P, A, 0, &imp_ms)) != MPLX_OK) return r;
  l->last_ms = lane->last_ms + imp_ms;
  return MPLX_OK;
}
// getSubStateSpace(k) = a Dijkstra from the k-th state of the last trajectory through the states that had been expanded in the
// space being left: the FILTER build of the speculative kernel with eps = 0 and no goal, then the same import.  P / A: lpa_params
// of the NEW space, A.old_* the old one.
static int lpa_subtree_by_import(mplx_lpa *l, int time_step, const SearchParams &P, const LpaParams &A) {
  int r;
  if ((r = lpa_import_lane(l, true)) != MPLX_OK) return r;
  mplx_ctx *lane = l->imp;
  mplx_waypoint start{};
  const double *s = &l->traj_states[(size_t)(l->traj_len - time_step) * 13];
  for (int k = 0; k < 3; k++) { start.pos[k] = s[k]; start.vel[k] = s[3 + k]; start.acc[k] = s[6 + k]; start.jrk[k] = s[9 + k]; }
  start.t = s[12];
  start.control = l->ctx->cfg.control;
  lane->filter_table = A.old_table; lane->filter_mask = A.old_table_mask; lane->filter_pool = A.old_node_pool; lane->filter_flag = FLAG_BUILT;
  mplx_result res;
  r = mplx_plan(lane, &start, &l->goal, &res);
  lane->filter_table = nullptr;
  if (r != MPLX_OK) return lfail(l, r, "%s", lane->err.c_str());
  float imp_ms = 0;
  return lpa_import_launch(l, P, A, 1, &imp_ms);
}

// getSubStateSpace(k) by planning afresh (choice L5b of DESIGN.md's LPA* section): an A* from the k-th state of the last trajectory to the
// planner's goal on the import lane, imported into the NEW space (P / A: lpa_params of it, A.old_*: the space being left).
static int lpa_subtree_by_fresh_plan(mplx_lpa *l, int time_step, const SearchParams &P, const LpaParams &A) {
  int r;
  if ((r = lpa_import_lane(l, false)) != MPLX_OK) return r;
  mplx_ctx *lane = l->imp;
  mplx_waypoint start{};
  const double *s = &l->traj_states[(size_t)(l->traj_len - time_step) * 13];
  for (int k = 0; k < 3; k++) { start.pos[k] = s[k]; start.vel[k] = s[3 + k]; start.acc[k] = s[6 + k]; start.jrk[k] = s[9 + k]; }
  start.t = s[12];
  start.control = l->ctx->cfg.control;
  mplx_result res;
  if ((r = mplx_plan(lane, &start, &l->goal, &res)) != MPLX_OK) return lfail(l, r, "%s", lane->err.c_str());
  float imp_ms = 0;
  return lpa_import_launch(l, P, A, 2, &imp_ms);
}

// PlannerBase::plan with setLPAstar(true) (map_replanner_node.cpp:141): repairs and re-uses the state space of the
// previous plan when the goal, the planner set-up and the start (= the current root) are unchanged; otherwise starts one.
extern "C" int mplx_lpa_plan(mplx_lpa *l, const mplx_waypoint *start, const mplx_waypoint *goal, mplx_result *out) {
  if (!l || !start || !goal || !out) return lfail(l, MPLX_ERR_ARG, "null argument");
  mplx_ctx *c = l->ctx;
  int r = check_ready(c);
  if (r) return lfail(l, r, "%s", c->err.c_str());
  if (start->enable_t) return lfail(l, MPLX_ERR_ARG, "enable_t is not supported by the voxel-map environment");
  if (!control_ok(goal->control)) return lfail(l, MPLX_ERR_ARG, "bad goal control");
  if (c->yaw) return lfail(l, MPLX_ERR_ARG, "LPA* over yaw-carrying states is not supported (no caller in the reference: map_replanner_node.cpp plans without yaw)");
  if (c->aux) return lfail(l, MPLX_ERR_ARG, "LPA* with a potential field / search region on the context is not supported (the edge cost would not be a function of the control input)");
  LCHK(l, hipSetDevice(c->device));
  if ((r = lpa_ensure(l)) != MPLX_OK) return r;
  QueryIn in{};
  mplx_waypoint s = *start;
  s.control = c->cfg.control;
  wp_to_state(s, in.start);
  wp_to_state(*goal, in.goal);
  in.start_t = start->t;
  in.goal_control = goal->control;
  int32_t key[MAX_KEY] = {};
  const int nk = state_key(c->cfg.control, in.start, key);
  bool fresh = !l->valid || !lpa_same_setup(l) || !lpa_same_goal(l->goal, *goal) || memcmp(key, l->root_key, sizeof(int32_t) * (size_t)nk) != 0;
  SearchParams P;
  LpaParams A;
  lpa_params(l, l->cur, P, A);
  A.fresh = fresh ? 1 : 0;
  static const bool no_import = getenv("MPLX_LPA_NO_IMPORT") != nullptr;  // (diagnostics: the fresh plan on the one-workgroup kernel, as before round 5)
  const bool by_import = fresh && !no_import && (c->cfg.control == CTRL_ACC || c->cfg.control == CTRL_JRK) && c->cfg.n_u <= 128 &&
                         (c->speculation < 0 || c->speculation > 1);
  if (by_import) {
    if ((r = lpa_plan_by_import(l, start, goal, P, A)) != MPLX_OK) {
      l->valid = false;
      l->traj_len = 0;
      return r;
    }
  } else {
    if (fresh) {
      LCHK(l, hipMemsetAsync(P.table, 0xFF, (size_t)l->table_slots * sizeof(unsigned long long), c->stream));
      LCHK(l, hipMemsetAsync(A.st, 0, sizeof(LpaState), c->stream));
    }
    LCHK(l, hipMemcpyAsync(l->d_in, &in, sizeof(QueryIn), hipMemcpyHostToDevice, c->stream));
    guard_arm(c);
    LCHK(l, hipEventRecord(c->ev0, c->stream));
    if (!mplx_launch_lpa(0, 0, c->stream, P, A)) return lfail(l, MPLX_ERR_ARG, "LPA* supports lattices of at most 128 control inputs (got %d)", P.n_u);
    LCHK(l, hipGetLastError());
    LCHK(l, hipEventRecord(c->ev1, c->stream));
    // (the guarded wait first: a device-to-host copy into pageable memory would block the host until the stream has drained)
    if (int rw = guard_wait(c, c->stream, "the LPA* search launch")) {  // aborted: the space was left in the middle of an expansion
      l->valid = false;
      l->traj_len = 0;
      return lfail(l, rw, "%s", c->err.c_str());
    }
    LCHK(l, hipStreamSynchronize(c->stream));
    LCHK(l, hipEventElapsedTime(&l->last_ms, c->ev0, c->ev1));
  }
  LCHK(l, hipMemcpyAsync(&l->last_out, l->d_out, sizeof(QueryOut), hipMemcpyDeviceToHost, c->stream));
  LCHK(l, hipMemcpyAsync(&l->st, A.st, sizeof(LpaState), hipMemcpyDeviceToHost, c->stream));
  LCHK(l, hipStreamSynchronize(c->stream));
  fill_result(l->last_out, *out);
  if (fresh) {  // whatever the outcome, the old space is gone
    l->valid = false;
    l->traj_len = 0;
  }
  // Pool exhaustion ends the launch in the middle of an expansion (the expanded state is already marked built, its
  // successors are not all linked): the space is not reusable -- the next plan() starts a fresh one, as after a failed
  // update (lpa_update).  The result of THIS call (status MPLX_PLAN_POOL_FULL) still says what happened.
  if (l->last_out.status == MPLX_PLAN_POOL_FULL) {
    l->valid = false;
    l->traj_len = 0;
  } else if (l->st.valid) {
    l->valid = true;
    l->cfg = c->cfg;
    l->U = c->U;
    l->cfg.U = nullptr;
    l->goal = *goal;
    if (fresh) memcpy(l->root_key, key, sizeof(key));
  }
  if (l->last_out.status == MPLX_PLAN_OK) {
    const int len = l->last_out.traj_len;
    l->traj_len = len;
    l->traj_nodes.assign((size_t)len + 1, 0);
    l->traj_actions.assign((size_t)std::max(len, 1), 0);
    l->traj_states.assign((size_t)(len + 1) * 13, 0.0);
    if (len > 0) {
      LCHK(l, hipMemcpyAsync(l->traj_nodes.data(), l->d_traj_nodes, sizeof(int32_t) * (size_t)(len + 1), hipMemcpyDeviceToHost, c->stream));
      LCHK(l, hipMemcpyAsync(l->traj_actions.data(), l->d_traj_actions, sizeof(int32_t) * (size_t)len, hipMemcpyDeviceToHost, c->stream));
      LCHK(l, hipMemcpyAsync(l->traj_states.data(), l->d_traj_states, sizeof(double) * (size_t)(len + 1) * 13, hipMemcpyDeviceToHost, c->stream));
      LCHK(l, hipStreamSynchronize(c->stream));
    }
  }
  return MPLX_OK;
}

static int lpa_update(mplx_lpa *l, int mode, int n_cells, const int32_t *cells, uint64_t *n_changed) {
  if (!l || n_cells < 0 || (n_cells > 0 && !cells)) return lfail(l, MPLX_ERR_ARG, "bad argument");
  if (n_changed) *n_changed = 0;
  if (!l->valid || n_cells == 0) return MPLX_OK;  // (replan_planner_.initialized() guards the calls upstream)
  mplx_ctx *c = l->ctx;
  if (!lpa_same_setup(l)) return lfail(l, MPLX_ERR_ARG, "the planner set-up on the context changed since the LPA* state space was built");
  LCHK(l, hipSetDevice(c->device));
  SearchParams P;
  LpaParams A;
  lpa_params(l, l->cur, P, A);
  // three passes (mplx_lpa.h): entries re-sampled by the whole machine, log entries converted in order by one thread,
  // the touched states' rhs recomputed by the whole machine
  LCHK(l, hipMemsetAsync(&A.st->n_changed, 0, sizeof(unsigned long long), c->stream));
  const int wide = std::max(1, 4 * c->n_cus);
  if (!mplx_launch_lpa(1, mode, c->stream, P, A, 0, wide)) return lfail(l, MPLX_ERR_ARG, "lattice too wide for LPA*");
  if (mode == 1) mplx_launch_lpa(1, mode, c->stream, P, A, 1, 1);
  mplx_launch_lpa(1, mode, c->stream, P, A, 2, wide);
  LCHK(l, hipGetLastError());
  LCHK(l, hipMemcpyAsync(&l->st, A.st, sizeof(LpaState), hipMemcpyDeviceToHost, c->stream));
  LCHK(l, hipStreamSynchronize(c->stream));
  if (l->st.n_changed == ~0ull) {
    l->valid = false;
    return lfail(l, MPLX_ERR_CAPACITY, "LPA* pools exhausted while turning cleared primitives into predecessor entries (mplx_lpa_set_capacity)");
  }
  if (n_changed) *n_changed = l->st.n_changed;
  return MPLX_OK;
}
// MapPlanner::updateBlockedNodes / updateClearedNodes (map_replanner_node.cpp:196,233), after the context's map was edited
extern "C" int mplx_lpa_update_blocked(mplx_lpa *l, int n_cells, const int32_t *cells, uint64_t *n_changed) { return lpa_update(l, 0, n_cells, cells, n_changed); }
extern "C" int mplx_lpa_update_cleared(mplx_lpa *l, int n_cells, const int32_t *cells, uint64_t *n_changed) { return lpa_update(l, 1, n_cells, cells, n_changed); }

constexpr uint32_t LPA_REROOT_AUTO = 16384;  // (the CPU checker of the tests uses the same constant: both sides must take the same branch)
extern "C" int mplx_lpa_set_reroot(mplx_lpa *l, int32_t mode) {
  if (!l || mode < 0 || mode > 2) return lfail(l, MPLX_ERR_ARG, "mode 0 (Dijkstra through the expanded states), 1 (plan afresh from the new root) or 2 (auto)");
  l->reroot_mode = mode;
  return MPLX_OK;
}
// PlannerBase::getSubStateSpace(time_step) (map_replanner_node.cpp:245)
extern "C" int mplx_lpa_sub_state_space(mplx_lpa *l, int32_t time_step) {
  if (!l) return MPLX_ERR_ARG;
  if (!l->valid || l->traj_len <= 0) return MPLX_OK;  // (best_child_ empty: nothing to do, like upstream)
  if (time_step < 0 || time_step > l->traj_len) return lfail(l, MPLX_ERR_ARG, "time_step %d outside the last trajectory (%d primitives)", time_step, l->traj_len);
  if (l->st.path[time_step] == NIL) return MPLX_OK;  // already re-rooted past that state
  mplx_ctx *c = l->ctx;
  if (!lpa_same_setup(l)) return lfail(l, MPLX_ERR_ARG, "the planner set-up on the context changed since the LPA* state space was built");
  LCHK(l, hipSetDevice(c->device));
  SearchParams P;
  LpaParams A, Aold;
  const int nxt = 1 - l->cur;
  {
    SearchParams Pold;
    lpa_params(l, l->cur, Pold, Aold);
    lpa_params(l, nxt, P, A);
    A.old_node_pool = Pold.node_pool;
    A.old_table = Pold.table;
    A.old_table_mask = Pold.table_mask;
    A.old_st = Aold.st;
    A.time_step = time_step;
  }
  static const bool no_import = getenv("MPLX_LPA_NO_IMPORT") != nullptr;  // (diagnostics: the one-workgroup Dijkstra, as before round 5)
  const bool by_import = !no_import && (c->cfg.control == CTRL_ACC || c->cfg.control == CTRL_JRK) && c->cfg.n_u <= 128 && (c->speculation < 0 || c->speculation > 1);
  LpaState ns{};
  // how (mplx_lpa_set_reroot; choices L5 / L5b of DESIGN.md's LPA* section): the Dijkstra through the expanded states of the space being
  // left, or -- cheaper as soon as that space is large: the new search has a heuristic and a goal -- an A* from the new root
  const bool fresh_plan = by_import && (l->reroot_mode == 1 || (l->reroot_mode == 2 && l->st.n_nodes > LPA_REROOT_AUTO));
  if (fresh_plan) {
    if (int ri = lpa_subtree_by_fresh_plan(l, time_step, P, A)) {
      l->valid = false;
      return ri;
    }
    LCHK(l, hipMemcpyAsync(&ns, A.st, sizeof(LpaState), hipMemcpyDeviceToHost, c->stream));
    LCHK(l, hipStreamSynchronize(c->stream));
    if (!ns.valid) {  // (the new root already satisfies the goal, or the pools ran out: no space is kept -- the next plan() starts one)
      const bool no_search = l->imp->last_out[0].n_nodes == 0;
      l->valid = false;
      if (no_search) return MPLX_OK;
      return lfail(l, MPLX_ERR_CAPACITY, "LPA* pools exhausted while rebuilding the sub state space (mplx_lpa_set_capacity)");
    }
  } else if (by_import) {
    if (int ri = lpa_subtree_by_import(l, time_step, P, A)) {
      l->valid = false;
      return ri;
    }
  } else {
    LCHK(l, hipMemsetAsync(P.table, 0xFF, (size_t)l->table_slots * sizeof(unsigned long long), c->stream));
    LCHK(l, hipMemsetAsync(A.st, 0, sizeof(LpaState), c->stream));
    guard_arm(c);
    if (!mplx_launch_lpa(2, 0, c->stream, P, A)) return lfail(l, MPLX_ERR_ARG, "lattice too wide for LPA*");
    LCHK(l, hipGetLastError());
    if (int rw = guard_wait(c, c->stream, "the LPA* sub-state-space launch")) {
      l->valid = false;
      return lfail(l, rw, "%s", c->err.c_str());
    }
  }
  LCHK(l, hipMemcpyAsync(&ns, A.st, sizeof(LpaState), hipMemcpyDeviceToHost, c->stream));
  LCHK(l, hipStreamSynchronize(c->stream));
  if (!ns.valid) {
    l->valid = false;
    return lfail(l, MPLX_ERR_CAPACITY, "LPA* pools exhausted while rebuilding the sub state space (mplx_lpa_set_capacity)");
  }
  l->st = ns;
  l->cur = nxt;
  // the new root's key, from the stored trajectory (device order is goal -> start)
  State rs{};
  const double *s = &l->traj_states[(size_t)(l->traj_len - time_step) * 13];
  for (int k = 0; k < 3; k++) { rs.p[k] = s[k]; rs.v[k] = s[3 + k]; rs.a[k] = s[6 + k]; rs.j[k] = s[9 + k]; }
  memset(l->root_key, 0, sizeof(l->root_key));
  state_key(l->cfg.control, rs, l->root_key);
  return MPLX_OK;
}

// getTraj() of the last successful plan (the stored Trajectory; unaffected by getSubStateSpace)
extern "C" int mplx_lpa_result_traj(mplx_lpa *l, mplx_primitive *prs, mplx_waypoint *wps, int32_t *actions, int32_t *node_ids) {
  if (!l) return MPLX_ERR_ARG;
  const int len = l->traj_len, control = l->cfg.control;
  for (int i = 0; i <= len && len > 0; i++) {
    const double *s = &l->traj_states[(size_t)(len - i) * 13];
    if (wps) {
      mplx_waypoint &w = wps[i];
      memset(&w, 0, sizeof(w));
      for (int k = 0; k < 3; k++) { w.pos[k] = s[k]; w.vel[k] = s[3 + k]; w.acc[k] = s[6 + k]; w.jrk[k] = s[9 + k]; }
      w.t = s[12];
      w.control = control;
    }
    if (node_ids) node_ids[i] = l->traj_nodes[(size_t)(len - i)];
  }
  for (int i = 0; i < len; i++) {
    const int a = l->traj_actions[(size_t)(len - 1 - i)];
    if (actions) actions[i] = a;
    if (prs) {
      const double *s = &l->traj_states[(size_t)(len - i) * 13];
      mplx_primitive &p = prs[i];
      memset(&p, 0, sizeof(p));
      for (int ax = 0; ax < 3; ax++) prim_build_axis(control, s[ax], s[3 + ax], s[6 + ax], s[9 + ax], l->U[3 * (size_t)a + ax], p.c[ax]);
      p.t = l->cfg.dt;
      p.control = control;
    }
  }
  return MPLX_OK;
}
extern "C" int mplx_lpa_traj_len(const mplx_lpa *l) { return l ? l->traj_len : 0; }
extern "C" int mplx_lpa_counts(const mplx_lpa *l, uint64_t *n_nodes, uint64_t *n_edges, uint64_t *n_blocked_log) {
  if (!l) return MPLX_ERR_ARG;
  if (n_nodes) *n_nodes = l->valid ? l->st.n_nodes : 0;
  if (n_edges) *n_edges = l->valid ? l->st.n_edges : 0;
  if (n_blocked_log) *n_blocked_log = l->valid ? l->st.n_blocked : 0;
  return MPLX_OK;
}
extern "C" int mplx_lpa_last_kernel_ms(const mplx_lpa *l, float *ms) {
  if (!l || !ms) return MPLX_ERR_ARG;
  *ms = l->last_ms;
  return MPLX_OK;
}
extern "C" int mplx_lpa_result_expanded(mplx_lpa *l, uint32_t cap, int32_t *ids, uint32_t *n) {
  if (!l || !ids || !n) return lfail(l, MPLX_ERR_ARG, "bad argument");
  if (!l->cap_rec || !l->d_rec) return lfail(l, MPLX_ERR_ARG, "recording disabled (mplx_lpa_set_record)");
  uint32_t cnt = std::min(l->last_out.n_recorded, cap);
  LCHK(l, hipSetDevice(l->ctx->device));
  if (cnt) LCHK(l, hipMemcpyAsync(ids, l->d_rec, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost, l->ctx->stream));
  LCHK(l, hipStreamSynchronize(l->ctx->stream));
  *n = cnt;
  return MPLX_OK;
}
// state-space dump: coords, g, rhs, h, closed / opened flags, `built` (expanded at least once); arrays of `cap` entries
extern "C" int mplx_lpa_result_nodes(mplx_lpa *l, uint64_t cap, mplx_waypoint *coords, double *g, double *rhs, double *h, int32_t *closed, int32_t *opened, int32_t *built) {
  if (!l) return MPLX_ERR_ARG;
  if (!l->valid) return MPLX_OK;
  const size_t n = l->st.n_nodes;
  if ((uint64_t)n > cap) return lfail(l, MPLX_ERR_CAPACITY, "state-space dump: %zu states, the caller's arrays hold %llu", n, (unsigned long long)cap);
  const int control = l->pool_control, nk = state_len(control), rb = rec_bytes(control), hot = rec_hot_bytes(control);
  LCHK(l, hipSetDevice(l->ctx->device));
  std::vector<char> buf(n * (size_t)rb);
  LCHK(l, hipMemcpyAsync(buf.data(), l->sp[l->cur].node_pool, buf.size(), hipMemcpyDeviceToHost, l->ctx->stream));
  LCHK(l, hipStreamSynchronize(l->ctx->stream));
  for (size_t i = 0; i < n; i++) {
    const char *r = buf.data() + i * rb;
    uint32_t fl;
    memcpy(&fl, r + 16, 4);
    if (g) memcpy(&g[i], r, 8);
    if (h) memcpy(&h[i], r + 8, 8);
    if (rhs) memcpy(&rhs[i], r + hot + (nk + 1) * 8, 8);
    if (closed) closed[i] = (fl & FLAG_CLOSED) ? 1 : 0;
    if (opened) opened[i] = (fl & FLAG_OPENED) ? 1 : 0;
    if (built) built[i] = (fl & FLAG_BUILT) ? 1 : 0;
    if (coords) {
      mplx_waypoint &w = coords[i];
      memset(&w, 0, sizeof(w));
      const double *st = (const double *)(r + hot);
      for (int d = 0; d < nk; d++) {
        double *dst = d < 3 ? w.pos : d < 6 ? w.vel : d < 9 ? w.acc : w.jrk;
        dst[d % 3] = st[d];
      }
      w.t = st[nk];
      w.control = control;
    }
  }
  return MPLX_OK;
}
// predecessor entries: for every state in id order its entries in arrival order; blocked: 1 when the entry's primitive
// is not free in the current map (cost +inf)
extern "C" int mplx_lpa_result_edges(mplx_lpa *l, int32_t *child, int32_t *parent, int32_t *action, int32_t *blocked, uint64_t cap, uint64_t *n_out) {
  if (!l || !n_out) return MPLX_ERR_ARG;
  *n_out = 0;
  if (!l->valid) return MPLX_OK;
  const size_t n_nodes = l->st.n_nodes, n_edges = l->st.n_edges;
  *n_out = n_edges;
  if (!n_nodes || !n_edges || cap == 0) return MPLX_OK;
  const int rb = rec_bytes(l->pool_control);
  LCHK(l, hipSetDevice(l->ctx->device));
  std::vector<char> buf(n_nodes * (size_t)rb);
  std::vector<EdgeRec> edges(n_edges);
  LCHK(l, hipMemcpyAsync(buf.data(), l->sp[l->cur].node_pool, buf.size(), hipMemcpyDeviceToHost, l->ctx->stream));
  LCHK(l, hipMemcpyAsync(edges.data(), l->sp[l->cur].edge_pool, n_edges * EDGE_BYTES, hipMemcpyDeviceToHost, l->ctx->stream));
  LCHK(l, hipStreamSynchronize(l->ctx->stream));
  uint64_t w = 0;
  std::vector<uint32_t> lst;
  for (size_t i = 0; i < n_nodes; i++) {
    uint32_t head;
    memcpy(&head, buf.data() + i * rb + 20, 4);
    lst.clear();
    for (uint32_t e = head; e != NIL; e = edges[e].next) {
      if (e >= n_edges || lst.size() > n_edges) return lfail(l, MPLX_ERR_ARG, "corrupt predecessor list of node %zu", i);
      lst.push_back(e);
    }
    for (size_t k = lst.size(); k-- > 0;) {
      const uint32_t e = lst[k];
      if (w < cap) {
        if (child) child[w] = (int32_t)i;
        if (parent) parent[w] = (int32_t)edges[e].parent;
        if (action) action[w] = (int32_t)(edges[e].action & ~EDGE_BLOCKED);
        if (blocked) blocked[w] = (edges[e].action & EDGE_BLOCKED) ? 1 : 0;
      }
      w++;
    }
  }
  *n_out = w;
  return MPLX_OK;
}
