// mplx_poly_lpa.hip -- C-ABI mplx_plpa_* (include/mplx.h): LPA* of the moving-obstacle planner -- PlannerBase::plan with
// setLPAstar(true), PolyMapPlanner::updateNodes, getSubStateSpace (poly_map_planner.h:61-93, poly_map_replanner_node.cpp:123-186,231).
// Kernels: mplx_poly_lpa.h.  A translation unit of its own; what it needs of the mplx_poly handle comes through
// mplx_poly_internal_view (mplx_poly_lpa_host.h).  No CPU fallback: every compute entry launches a gfx950 kernel or fails.
#include "../../include/mplx.h"

#include <hip/hip_runtime.h>

#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "mplx_poly_lpa.h"
#include "mplx_poly_lpa_host.h"

using namespace mplx;

extern "C" const char *mplx_poly_last_error(const mplx_poly *p);

struct mplx_plpa {
  mplx_poly *poly = nullptr;
  std::string err;
  // capacities and pools (flat, private to the handle: chunk tables are the identity)
  uint64_t cap_nodes = 1 << 18, cap_edges = 1 << 21, cap_log = 1 << 21;
  bool pools_valid = false;
  int pool_control = 0;
  char *node_pool = nullptr, *edge_pool = nullptr, *open_pool = nullptr;
  unsigned long long *table = nullptr;
  uint64_t table_slots = 0;
  uint32_t *bkt_head = nullptr;
  LpaState *d_st = nullptr;
  QueryIn *d_in = nullptr;
  QueryOut *d_out = nullptr;
  int32_t *d_traj_nodes = nullptr, *d_traj_actions = nullptr, *d_rec = nullptr;
  double *d_traj_states = nullptr;
  uint32_t *d_changed = nullptr, *d_counters = nullptr;
  double *d_edge_cost = nullptr;
  uint32_t *d_succ_child = nullptr, *d_succ_entry = nullptr;  // per state x control input (null when that would be too large)
  int succ_n_u = 0;
  uint64_t synced_epoch = ~0ull;  // mplx_poly commit count the entries' blocked bits were last brought in step with
  uint32_t cap_rec = 0;
  // host copies
  LpaState st{};
  QueryOut last_out{};
  bool valid = false;
  double goal[9] = {0}, eps = 1.0, tol_pos = 0.5, tol_vel = -1.0;
  int32_t max_expand = -1, heur_ignore_dynamics = 1;
  int32_t root_key[MAX_KEY + 1] = {0};
  int traj_len = 0;
  std::vector<int32_t> traj_nodes, traj_actions;
  std::vector<double> traj_states;
  std::vector<uint32_t> changed;
  float last_ms = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

static int lf(mplx_plpa *l, int code, const char *fmt, ...) {
  char buf[768];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (l) l->err = buf;
  return code;
}
#define LH(l, x)                                                                                       \
  do {                                                                                                 \
    hipError_t e_ = (x);                                                                               \
    if (e_ != hipSuccess) return lf(l, MPLX_ERR_HIP, "%s failed: %s", #x, hipGetErrorString(e_));      \
  } while (0)

static uint64_t np2(uint64_t v) {
  uint64_t p = 1;
  while (p < v) p <<= 1;
  return p;
}
static void plpa_free(mplx_plpa *l) {
  (void)hipFree(l->node_pool); (void)hipFree(l->edge_pool); (void)hipFree(l->open_pool); (void)hipFree(l->table); (void)hipFree(l->bkt_head);
  (void)hipFree(l->d_st); (void)hipFree(l->d_in); (void)hipFree(l->d_out); (void)hipFree(l->d_traj_nodes); (void)hipFree(l->d_traj_actions);
  (void)hipFree(l->d_traj_states); (void)hipFree(l->d_changed); (void)hipFree(l->d_counters); (void)hipFree(l->d_rec); (void)hipFree(l->d_edge_cost); (void)hipFree(l->d_succ_child); (void)hipFree(l->d_succ_entry);
  l->d_edge_cost = nullptr; l->d_succ_child = l->d_succ_entry = nullptr;
  l->node_pool = l->edge_pool = l->open_pool = nullptr;
  l->table = nullptr; l->bkt_head = nullptr; l->d_st = nullptr; l->d_in = nullptr; l->d_out = nullptr;
  l->d_traj_nodes = l->d_traj_actions = l->d_rec = nullptr; l->d_traj_states = nullptr; l->d_changed = l->d_counters = nullptr;
  l->pools_valid = false;
}
static int plpa_ensure(mplx_plpa *l, int control, int n_u) {
  if (l->pools_valid && l->pool_control == control && l->succ_n_u == n_u) return MPLX_OK;
  plpa_free(l);
  l->valid = false;
  const uint64_t nch = std::max<uint64_t>(1, (l->cap_nodes + (1u << NODE_CH_LOG) - 1) >> NODE_CH_LOG);
  const uint64_t ech = std::max<uint64_t>(1, (l->cap_edges + (1u << EDGE_CH_LOG) - 1) >> EDGE_CH_LOG);
  const uint64_t och = std::max<uint64_t>(1, (l->cap_log + (1u << OPEN_CH_LOG) - 1) >> OPEN_CH_LOG);
  if (nch > (uint64_t)MAX_NODE_CH || ech > (uint64_t)MAX_EDGE_CH || och > (uint64_t)MAX_OPEN_CH)
    return lf(l, MPLX_ERR_ARG, "capacity too large (at most %d node / %d predecessor / %d OPEN-log chunks)", MAX_NODE_CH, MAX_EDGE_CH, MAX_OPEN_CH);
  l->table_slots = np2(4ull * (nch << NODE_CH_LOG));
  LH(l, hipMalloc((void **)&l->node_pool, (size_t)(nch << NODE_CH_LOG) * rec_bytes(control)));
  LH(l, hipMalloc((void **)&l->edge_pool, (size_t)(ech << EDGE_CH_LOG) * EDGE_BYTES));
  LH(l, hipMalloc((void **)&l->open_pool, (size_t)(och << OPEN_CH_LOG) * OPEN_BYTES));
  LH(l, hipMalloc((void **)&l->table, (size_t)l->table_slots * sizeof(unsigned long long)));
  LH(l, hipMalloc((void **)&l->bkt_head, sizeof(uint32_t) * 2 * NB * NSUB));
  LH(l, hipMemset(l->bkt_head, 0xFF, sizeof(uint32_t) * 2 * NB * NSUB));
  LH(l, hipMalloc((void **)&l->d_st, sizeof(LpaState)));
  LH(l, hipMalloc((void **)&l->d_in, sizeof(QueryIn)));
  LH(l, hipMalloc((void **)&l->d_out, sizeof(QueryOut)));
  LH(l, hipMalloc((void **)&l->d_traj_nodes, sizeof(int32_t) * (MAX_TRAJ + 1)));
  LH(l, hipMalloc((void **)&l->d_traj_actions, sizeof(int32_t) * MAX_TRAJ));
  LH(l, hipMalloc((void **)&l->d_traj_states, sizeof(double) * (MAX_TRAJ + 1) * 13));
  LH(l, hipMalloc((void **)&l->d_changed, sizeof(uint32_t) * (size_t)(ech << EDGE_CH_LOG)));
  LH(l, hipMalloc((void **)&l->d_counters, sizeof(uint32_t) * 4));
  LH(l, hipMalloc((void **)&l->d_edge_cost, sizeof(double) * (size_t)(ech << EDGE_CH_LOG)));
  l->succ_n_u = n_u;
  if ((nch << NODE_CH_LOG) * (uint64_t)n_u * 8ull <= (4ull << 30)) {  // (beyond 4 GB the re-expansion shortcut is off: get_succ runs again)
    LH(l, hipMalloc((void **)&l->d_succ_child, sizeof(uint32_t) * (size_t)(nch << NODE_CH_LOG) * (size_t)n_u));
    LH(l, hipMalloc((void **)&l->d_succ_entry, sizeof(uint32_t) * (size_t)(nch << NODE_CH_LOG) * (size_t)n_u));
  }
  l->cap_rec = (uint32_t)std::min<uint64_t>(l->cap_nodes, 1u << 24);
  LH(l, hipMalloc((void **)&l->d_rec, sizeof(int32_t) * (size_t)l->cap_rec));
  l->pool_control = control;
  l->pools_valid = true;
  return MPLX_OK;
}
static void plpa_params(const mplx_plpa *l, const mplx_poly_view &v, SearchParams &P) {
  P = SearchParams{};
  P.control = v.dev.control; P.n_u = v.dev.n_u;
  P.ns = state_len(v.dev.control); P.nk = state_len(v.dev.control);
  P.dt = v.dev.dt; P.v_max = v.dev.v_max; P.a_max = v.dev.a_max; P.j_max = v.dev.j_max; P.w = v.dev.w;
  P.eps = l->eps; P.tol_pos = l->tol_pos; P.tol_vel = l->tol_vel; P.tol_acc = -1.0; P.t_max = INFINITY;
  P.max_expand = l->max_expand; P.heur_ignore_dynamics = l->heur_ignore_dynamics;
  // (coarse OPEN bucket; a fine one is 1 / 1024 of it.  Measurement: MPLX_PLPA_BUCKET_FACTOR)
  static const double bucket_factor = [] { const char *e = getenv("MPLX_PLPA_BUCKET_FACTOR"); const double v = e ? atof(e) : 0.0; return v > 0 ? v : 8.0; }();
  P.bucket_width = P.w * P.dt > 0 ? P.w * P.dt * bucket_factor : 1.0;
  P.node_pool = l->node_pool; P.edge_pool = l->edge_pool; P.open_pool = l->open_pool;
  P.node_chunks = (uint32_t)std::max<uint64_t>(1, (l->cap_nodes + (1u << NODE_CH_LOG) - 1) >> NODE_CH_LOG);
  P.edge_chunks = (uint32_t)std::max<uint64_t>(1, (l->cap_edges + (1u << EDGE_CH_LOG) - 1) >> EDGE_CH_LOG);
  P.open_chunks = (uint32_t)std::max<uint64_t>(1, (l->cap_log + (1u << OPEN_CH_LOG) - 1) >> OPEN_CH_LOG);
  P.table = l->table; P.table_mask = l->table_slots - 1;
  P.bkt_head = l->bkt_head;
  P.cap_rec = l->cap_rec;
  P.nq = 1;
  P.queries = l->d_in; P.out = l->d_out;
  P.traj_nodes = l->d_traj_nodes; P.traj_actions = l->d_traj_actions; P.traj_states = l->d_traj_states;
  P.rec_ids = l->d_rec;
  P.guard = v.guard;
  P.poly = v.dev;
}
template <bool GEN>
static void launch_plan(int control, hipStream_t s, const SearchParams &P, const PlpaArgs &A) {
  if (control == CTRL_ACC) hipLaunchKernelGGL((plpa_plan_kernel<CTRL_ACC, GEN>), dim3(1), dim3(64), 0, s, P, A);
  else hipLaunchKernelGGL((plpa_plan_kernel<CTRL_JRK, GEN>), dim3(1), dim3(64), 0, s, P, A);
}
template <bool GEN>
static void launch_update(int control, int grid, hipStream_t s, const SearchParams &P, const PlpaArgs &A) {
  if (control == CTRL_ACC) hipLaunchKernelGGL((plpa_update_kernel<CTRL_ACC, GEN>), dim3(grid), dim3(64), 0, s, P, A);
  else hipLaunchKernelGGL((plpa_update_kernel<CTRL_JRK, GEN>), dim3(grid), dim3(64), 0, s, P, A);
}
// wait for the stream with the handle's launch deadline (opt-in, counted from `t0`): past it the abort word of the planner context's
// guard block is raised, which plpa_plan_kernel polls
static int plpa_wait(mplx_plpa *l, const mplx_poly_view &v, std::chrono::steady_clock::time_point t0, const char *what) {
  using clk = std::chrono::steady_clock;
  bool aborted = false;
  for (;;) {
    const hipError_t e = hipStreamQuery(v.stream);
    if (e == hipSuccess) break;
    if (e != hipErrorNotReady) return lf(l, MPLX_ERR_HIP, "hipStreamQuery failed while waiting for %s: %s", what, hipGetErrorString(e));
    const double el = std::chrono::duration<double>(clk::now() - t0).count();
    if (v.deadline_s > 0 && !aborted && el > v.deadline_s && v.guard) {
      __atomic_store_n(&v.guard->abort, 1u, __ATOMIC_SEQ_CST);
      aborted = true;
    }
    if (aborted && el > v.deadline_s + 10.0) return lf(l, MPLX_ERR_TIMEOUT, "%s did not end within %.1f s of its launch and did not answer the abort word", what, v.deadline_s);
    usleep(el < 0.05 ? 50 : 250);
  }
  if (aborted) {
    __atomic_store_n(&v.guard->abort, 0u, __ATOMIC_SEQ_CST);
    return lf(l, MPLX_ERR_TIMEOUT, "%s did not end within %.1f s of its launch and was aborted (results of the launch are void)", what, v.deadline_s);
  }
  return MPLX_OK;
}

extern "C" int mplx_plpa_create(mplx_poly *poly, mplx_plpa **out) {
  if (!poly || !out) return MPLX_ERR_ARG;
  mplx_plpa *l = new mplx_plpa();
  l->poly = poly;
  *out = l;
  return MPLX_OK;
}
extern "C" void mplx_plpa_destroy(mplx_plpa *l) {
  if (!l) return;
  plpa_free(l);
  if (l->ev0) (void)hipEventDestroy(l->ev0);
  if (l->ev1) (void)hipEventDestroy(l->ev1);
  delete l;
}
extern "C" const char *mplx_plpa_last_error(const mplx_plpa *l) { return l ? l->err.c_str() : ""; }
extern "C" int mplx_plpa_set_capacity(mplx_plpa *l, uint64_t max_nodes, uint64_t max_edges, uint64_t max_open_log) {
  if (!l) return MPLX_ERR_ARG;
  if (max_nodes) l->cap_nodes = max_nodes;
  if (max_edges) l->cap_edges = max_edges;
  if (max_open_log) l->cap_log = max_open_log;
  l->pools_valid = false;
  return MPLX_OK;
}
extern "C" int mplx_plpa_initialized(const mplx_plpa *l) { return l && l->valid ? 1 : 0; }
extern "C" int mplx_plpa_reset(mplx_plpa *l) {
  if (!l) return MPLX_ERR_ARG;
  l->valid = false;
  l->traj_len = 0;
  return MPLX_OK;
}

static int plpa_view(mplx_plpa *l, int32_t world, mplx_poly_view &v) {
  const int r = mplx_poly_internal_view(l->poly, &v);
  if (r != MPLX_OK) return lf(l, r, "%s", mplx_poly_last_error(l->poly));
  if (v.dev.control != CTRL_ACC && v.dev.control != CTRL_JRK) return lf(l, MPLX_ERR_ARG, "the moving-obstacle LPA* runs ACC or JRK states");
  if (v.dev.n_u > POLY_MAX_U) return lf(l, MPLX_ERR_ARG, "at most %d control inputs", POLY_MAX_U);
  if (world < 0 || world >= v.n_worlds) return lf(l, MPLX_ERR_ARG, "world index out of range");
  return MPLX_OK;
}
static void key_of(int control, const double *s9, int32_t *key) {
  State st{};
  st.p[0] = s9[0]; st.p[1] = s9[1]; st.v[0] = s9[2]; st.v[1] = s9[3];
  if (control == CTRL_JRK) { st.a[0] = s9[4]; st.a[1] = s9[5]; }
  const int n = state_key(control, st, key);
  key[n] = (int32_t)round(s9[8] / 0.1);
}

// PlannerBase::plan with setLPAstar(true) (poly_map_replanner_node.cpp:141): repairs and re-uses the state space of the previous plan when
// the goal and the start (= the current root) are unchanged (L6); otherwise starts one.  start / goal: pos2 vel2 acc2 jrk2 t.
static int plpa_plan_impl(mplx_plpa *l, int32_t world, const double *start, const double *goal, bool force_fresh, mplx_result *out) {
  mplx_poly_view v;
  int r = plpa_view(l, world, v);
  if (r) return r;
  LH(l, hipSetDevice(v.device));
  const int control = v.dev.control;
  if ((r = plpa_ensure(l, control, v.dev.n_u)) != MPLX_OK) return r;
  if (!l->ev0) { LH(l, hipEventCreate(&l->ev0)); LH(l, hipEventCreate(&l->ev1)); }
  QueryIn in{};
  in.start.p[0] = start[0]; in.start.p[1] = start[1]; in.start.v[0] = start[2]; in.start.v[1] = start[3];
  in.goal.p[0] = goal[0]; in.goal.p[1] = goal[1]; in.goal.v[0] = goal[2]; in.goal.v[1] = goal[3];
  if (control == CTRL_JRK) { in.start.a[0] = start[4]; in.start.a[1] = start[5]; in.goal.a[0] = goal[4]; in.goal.a[1] = goal[5]; }
  in.start_t = start[8];
  in.goal_control = control;
  int32_t key[MAX_KEY + 1] = {0};
  key_of(control, start, key);
  bool same_goal = true;
  for (int i = 0; i < 8; i++) same_goal = same_goal && l->goal[i] == goal[i];
  const bool fresh = force_fresh || !l->valid || !same_goal || memcmp(key, l->root_key, sizeof(key)) != 0;
  SearchParams P;
  plpa_params(l, v, P);
  PlpaArgs A{};
  A.st = l->d_st; A.fresh = fresh ? 1 : 0; A.world = world; A.edge_cost = l->d_edge_cost;
  A.succ_child = l->d_succ_child; A.succ_entry = l->d_succ_entry;
  // the entries are in step with the committed world when the space is new, or when updateNodes ran after the last commit
  if (fresh) l->synced_epoch = v.commit_epoch;
  A.trust_entries = (l->synced_epoch == v.commit_epoch && getenv("MPLX_PLPA_NO_REUSE") == nullptr) ? 1 : 0;
  hipStream_t s = v.stream;
  LH(l, hipMemcpyAsync(l->d_in, &in, sizeof(QueryIn), hipMemcpyHostToDevice, s));
  if (fresh) {
    LH(l, hipMemsetAsync(l->table, 0xFF, (size_t)l->table_slots * sizeof(unsigned long long), s));
    LH(l, hipMemsetAsync(l->d_st, 0, sizeof(LpaState), s));
  }
  if (v.guard) memset(v.guard, 0, sizeof(GuardBlock));
  const auto t0 = std::chrono::steady_clock::now();
  LH(l, hipEventRecord(l->ev0, s));
  if (v.general) launch_plan<true>(control, s, P, A); else launch_plan<false>(control, s, P, A);
  LH(l, hipGetLastError());
  LH(l, hipEventRecord(l->ev1, s));
  if ((r = plpa_wait(l, v, t0, "the moving-obstacle LPA* launch")) != MPLX_OK) {
    l->valid = false;
    return r;
  }
  LH(l, hipMemcpyAsync(&l->last_out, l->d_out, sizeof(QueryOut), hipMemcpyDeviceToHost, s));
  LH(l, hipMemcpyAsync(&l->st, l->d_st, sizeof(LpaState), hipMemcpyDeviceToHost, s));
  LH(l, hipStreamSynchronize(s));
  LH(l, hipEventElapsedTime(&l->last_ms, l->ev0, l->ev1));
  const QueryOut &o = l->last_out;
  if (out) {
    memset(out, 0, sizeof(*out));
    out->status = o.status; out->traj_len = o.traj_len; out->cost = o.cost;
    out->n_expanded = o.n_expanded; out->n_closed = o.n_closed; out->n_nodes = o.n_nodes; out->n_edges = o.n_edges;
    out->n_primitives = o.n_primitives; out->n_succ = o.n_succ; out->n_succ_finite = o.n_succ_finite;
    out->n_push = o.n_push; out->n_refill = o.n_refill; out->n_evict = o.n_evict; out->expand_hash = o.expand_hash;
  }
  if (o.status == MPLX_PLAN_POOL_FULL || o.status == MPLX_PLAN_INTERNAL) {
    l->valid = false;
    l->traj_len = 0;
    if (o.status == MPLX_PLAN_INTERNAL) return lf(l, MPLX_ERR_ARG, "internal: a hyperplane equation of unsupported degree was met, or the search ran away");
  } else if (l->st.valid) {
    l->valid = true;
    for (int i = 0; i < 9; i++) l->goal[i] = goal[i];
    if (fresh) memcpy(l->root_key, key, sizeof(key));
  } else if (fresh) {
    l->valid = false;  // (the start already satisfies the goal, or lies outside the map: no state space was built)
    for (int i = 0; i < 9; i++) l->goal[i] = goal[i];
  }
  l->traj_len = 0;
  if (o.status == MPLX_PLAN_OK && o.traj_len > 0) {
    const int len = o.traj_len;
    l->traj_len = len;
    l->traj_nodes.assign((size_t)len + 1, 0);
    l->traj_actions.assign((size_t)len, 0);
    l->traj_states.assign((size_t)(len + 1) * 13, 0.0);
    LH(l, hipMemcpy(l->traj_nodes.data(), l->d_traj_nodes, sizeof(int32_t) * (size_t)(len + 1), hipMemcpyDeviceToHost));
    LH(l, hipMemcpy(l->traj_actions.data(), l->d_traj_actions, sizeof(int32_t) * (size_t)len, hipMemcpyDeviceToHost));
    LH(l, hipMemcpy(l->traj_states.data(), l->d_traj_states, sizeof(double) * (size_t)(len + 1) * 13, hipMemcpyDeviceToHost));
  }
  return MPLX_OK;
}
extern "C" int mplx_plpa_plan(mplx_plpa *l, int32_t world, const double *start, const double *goal, double eps, double tol_pos, double tol_vel, int32_t max_expand,
                              int32_t heur_ignore_dynamics, mplx_result *out) {
  if (!l || !start || !goal || !out) return lf(l, MPLX_ERR_ARG, "null argument");
  if (l->valid && (l->eps != eps || l->tol_pos != tol_pos || l->tol_vel != tol_vel || l->heur_ignore_dynamics != heur_ignore_dynamics)) l->valid = false;
  l->eps = eps; l->tol_pos = tol_pos; l->tol_vel = tol_vel; l->max_expand = max_expand; l->heur_ignore_dynamics = heur_ignore_dynamics;
  return plpa_plan_impl(l, world, start, goal, false, out);
}

// PolyMapPlanner::updateNodes (poly_map_planner.h:61-93), after the world's obstacles / start time were committed again
extern "C" int mplx_plpa_update_nodes(mplx_plpa *l, int32_t world, uint64_t *n_blocked, uint64_t *n_cleared) {
  if (!l) return MPLX_ERR_ARG;
  if (n_blocked) *n_blocked = 0;
  if (n_cleared) *n_cleared = 0;
  l->changed.clear();
  if (!l->valid) return MPLX_OK;  // (updateNodes returns at once without a state space: poly_map_planner.h:65)
  mplx_poly_view v;
  int r = plpa_view(l, world, v);
  if (r) return r;
  if (!l->pools_valid || l->pool_control != v.dev.control) return lf(l, MPLX_ERR_ARG, "the planner set-up changed since the state space was built");
  LH(l, hipSetDevice(v.device));
  SearchParams P;
  plpa_params(l, v, P);
  PlpaArgs A{};
  A.st = l->d_st; A.world = world; A.changed = l->d_changed; A.counters = l->d_counters; A.edge_cost = l->d_edge_cost;
  A.succ_child = l->d_succ_child; A.succ_entry = l->d_succ_entry;
  A.changed_cap = (uint32_t)std::min<uint64_t>((uint64_t)P.edge_chunks << EDGE_CH_LOG, 0xFFFFFFF0ull);
  hipStream_t s = v.stream;
  LH(l, hipMemsetAsync(l->d_counters, 0, sizeof(uint32_t) * 4, s));
  const int grid = (int)std::min<uint64_t>(1024, (l->st.n_nodes + 63) / 64 + 1);
  if (v.general) launch_update<true>(v.dev.control, grid, s, P, A); else launch_update<false>(v.dev.control, grid, s, P, A);
  LH(l, hipGetLastError());
  uint32_t ctr[4] = {0, 0, 0, 0};
  LH(l, hipMemcpyAsync(ctr, l->d_counters, sizeof(ctr), hipMemcpyDeviceToHost, s));
  LH(l, hipStreamSynchronize(s));
  if (ctr[3]) { l->valid = false; return lf(l, MPLX_ERR_ARG, "internal: a hyperplane equation of unsupported degree was met"); }
  l->synced_epoch = v.commit_epoch;  // every entry has just been re-tested against the world as committed now
  const uint32_t n = std::min(ctr[2], A.changed_cap);
  l->changed.resize(n);
  if (n) LH(l, hipMemcpy(l->changed.data(), l->d_changed, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
  std::sort(l->changed.begin(), l->changed.end(), [](uint32_t a, uint32_t b) { return (a & 0x7FFFFFFFu) < (b & 0x7FFFFFFFu); });
  if (n_blocked) *n_blocked = ctr[0];
  if (n_cleared) *n_cleared = ctr[1];
  return MPLX_OK;
}
// the entries updateNodes changed, by entry number: what getBlockedPrimitives / getClearedPrimitives are built from (entry -> parent state
// and action: mplx_plpa_result_entries / _nodes)
extern "C" int mplx_plpa_changed(mplx_plpa *l, uint64_t cap, int32_t *entry, int32_t *now_blocked, uint64_t *n) {
  if (!l || !n) return MPLX_ERR_ARG;
  *n = l->changed.size();
  for (size_t i = 0; i < l->changed.size() && i < cap; i++) {
    if (entry) entry[i] = (int32_t)(l->changed[i] & 0x7FFFFFFFu);
    if (now_blocked) now_blocked[i] = (int32_t)(l->changed[i] >> 31);
  }
  return MPLX_OK;
}
// PlannerBase::getSubStateSpace(time_step) (poly_map_replanner_node.cpp:231): by planning afresh from the time_step-th state of the
// last trajectory to the planner's goal (L5b); the stored trajectory is dropped (the caller plans from its state next)
extern "C" int mplx_plpa_sub_state_space(mplx_plpa *l, int32_t world, int32_t time_step) {
  if (!l) return MPLX_ERR_ARG;
  if (!l->valid || l->traj_len <= 0) return MPLX_OK;
  if (time_step < 0 || time_step > l->traj_len) return lf(l, MPLX_ERR_ARG, "time_step %d outside the last trajectory (%d primitives)", time_step, l->traj_len);
  const double *s = &l->traj_states[(size_t)(l->traj_len - time_step) * 13];  // (device order is goal -> start)
  double start[9] = {s[0], s[1], s[3], s[4], s[6], s[7], 0.0, 0.0, s[12]};
  double goal[9];
  for (int i = 0; i < 9; i++) goal[i] = l->goal[i];
  mplx_result res;
  const int r = plpa_plan_impl(l, world, start, goal, true, &res);
  l->traj_len = 0;
  return r;
}
extern "C" int mplx_plpa_traj_len(const mplx_plpa *l) { return l ? l->traj_len : 0; }
// trajectory of the last successful plan, start -> goal: actions[len], node_ids[len + 1], states (len + 1) x 9 (pos2 vel2 acc2 jrk2 t)
extern "C" int mplx_plpa_result_traj(mplx_plpa *l, int32_t *actions, int32_t *node_ids, double *states) {
  if (!l) return MPLX_ERR_ARG;
  const int len = l->traj_len;
  for (int i = 0; i <= len && len > 0; i++) {
    const double *s = &l->traj_states[(size_t)(len - i) * 13];
    if (node_ids) node_ids[i] = l->traj_nodes[(size_t)(len - i)];
    if (states) {
      double *o = states + 9 * (size_t)i;
      o[0] = s[0]; o[1] = s[1]; o[2] = s[3]; o[3] = s[4]; o[4] = s[6]; o[5] = s[7]; o[6] = 0.0; o[7] = 0.0; o[8] = s[12];
    }
  }
  for (int i = 0; i < len; i++)
    if (actions) actions[i] = l->traj_actions[(size_t)(len - 1 - i)];
  return MPLX_OK;
}
extern "C" int mplx_plpa_last_kernel_ms(const mplx_plpa *l, float *ms) {
  if (!l || !ms) return MPLX_ERR_ARG;
  *ms = l->last_ms;
  return MPLX_OK;
}
// thread 0's clock per section of the last plan's iterations (shader cycles): 0 pop, 1 stop test + settling, 2 primitives / look-ups /
// heuristics, 3 isFree of the primitives, 4 link, 5 updateNode of the children (look-ahead values, flags, pushes), 6 goal test + barrier
extern "C" int mplx_plpa_result_cycles(const mplx_plpa *l, uint64_t cyc[10]) {
  if (!l || !cyc) return MPLX_ERR_ARG;
  for (int i = 0; i < 10; i++) cyc[i] = l->last_out.cyc[i];
  return MPLX_OK;
}
extern "C" int mplx_plpa_counts(const mplx_plpa *l, uint64_t *n_nodes, uint64_t *n_entries) {
  if (!l) return MPLX_ERR_ARG;
  if (n_nodes) *n_nodes = l->valid ? l->st.n_nodes : 0;
  if (n_entries) *n_entries = l->valid ? l->st.n_edges : 0;
  return MPLX_OK;
}
extern "C" int mplx_plpa_result_expanded(mplx_plpa *l, uint32_t cap, int32_t *ids, uint32_t *n) {
  if (!l || !ids || !n) return MPLX_ERR_ARG;
  const uint32_t cnt = std::min(l->last_out.n_recorded, cap);
  if (cnt) LH(l, hipMemcpy(ids, l->d_rec, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost));
  *n = cnt;
  return MPLX_OK;
}
// state-space dump: per state pos2 vel2 acc2 jrk2 t | g rhs h | closed opened built; arrays of `cap` states
extern "C" int mplx_plpa_result_nodes(mplx_plpa *l, uint64_t cap, double *states, double *g, double *rhs, double *h, int32_t *closed, int32_t *opened, int32_t *built) {
  if (!l) return MPLX_ERR_ARG;
  if (!l->valid) return MPLX_OK;
  const size_t n = l->st.n_nodes;
  if ((uint64_t)n > cap) return lf(l, MPLX_ERR_CAPACITY, "state-space dump: %zu states, the caller's arrays hold %llu", n, (unsigned long long)cap);
  const int control = l->pool_control, ns = state_len(control), rb = rec_bytes(control), hot = rec_hot_bytes(control);
  std::vector<char> buf(n * (size_t)rb);
  LH(l, hipMemcpy(buf.data(), l->node_pool, buf.size(), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; i++) {
    const char *r = buf.data() + i * rb;
    const double *st = (const double *)(r + hot);
    if (states) {
      double *o = states + 9 * i;
      o[0] = st[0]; o[1] = st[1]; o[2] = st[3]; o[3] = st[4]; o[4] = ns > 6 ? st[6] : 0.0; o[5] = ns > 6 ? st[7] : 0.0; o[6] = 0.0; o[7] = 0.0; o[8] = st[ns];
    }
    const uint32_t fl = *(const uint32_t *)(r + 16);
    if (g) g[i] = *(const double *)r;
    if (h) h[i] = *(const double *)(r + 8);
    if (rhs) rhs[i] = st[ns + 1];
    if (closed) closed[i] = (fl & FLAG_CLOSED) ? 1 : 0;
    if (opened) opened[i] = (fl & FLAG_OPENED) ? 1 : 0;
    if (built) built[i] = (fl & FLAG_BUILT) ? 1 : 0;
  }
  return MPLX_OK;
}
// predecessor entries in creation order: child, parent, action, blocked; arrays of `cap` entries
extern "C" int mplx_plpa_result_entries(mplx_plpa *l, uint64_t cap, int32_t *child, int32_t *parent, int32_t *action, int32_t *blocked) {
  if (!l) return MPLX_ERR_ARG;
  if (!l->valid) return MPLX_OK;
  const size_t n = l->st.n_nodes, ne = l->st.n_edges;
  if ((uint64_t)ne > cap) return lf(l, MPLX_ERR_CAPACITY, "entry dump: %zu entries, the caller's arrays hold %llu", ne, (unsigned long long)cap);
  const int rb = rec_bytes(l->pool_control);
  std::vector<char> nb(n * (size_t)rb);
  std::vector<EdgeRec> eb(ne);
  LH(l, hipMemcpy(nb.data(), l->node_pool, nb.size(), hipMemcpyDeviceToHost));
  if (ne) LH(l, hipMemcpy(eb.data(), l->edge_pool, sizeof(EdgeRec) * ne, hipMemcpyDeviceToHost));
  for (size_t e = 0; e < ne; e++) {
    if (parent) parent[e] = (int32_t)eb[e].parent;
    if (action) action[e] = (int32_t)(eb[e].action & ~EDGE_BLOCKED);
    if (blocked) blocked[e] = (eb[e].action & EDGE_BLOCKED) ? 1 : 0;
  }
  if (child)
    for (size_t i = 0; i < n; i++)
      for (uint32_t e = *(const uint32_t *)(nb.data() + i * rb + 20); e != NIL && e < ne; e = eb[e].next) child[e] = (int32_t)i;
  return MPLX_OK;
}
