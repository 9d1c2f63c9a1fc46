// mplx_lpa.h -- LPA* (incremental replanning) on a device-resident state space that persists across plan() calls.
//
// What the reference's replanner does with it (mpl_test_node/src/map_replanner_node.cpp): setLPAstar(true) (:437), plan()
// (:141), then, whenever the shared MapUtil was edited, updateBlockedNodes(new_obs) / updateClearedNodes(new_clear)
// (:196, :233) and plan() again; getSubStateSpace(1) + start = getTraj().getWaypoints()[1] to move on (:243-255).  The
// repair mechanism is in-tree for the sibling planner (poly_map_planner.h:61-93): re-evaluate every stored predecessor
// entry with forward_action + isFree, increaseCost what became blocked, decreaseCost what became free.  The search
// itself (graph_search.h LPAstar, state_space.h) is un-vendored; the algorithm is Koenig & Likhachev's LPA* and the
// choices L1-L6 listed in DESIGN.md (LPA* section; the CPU checker of the tests states the same six) apply here:
//   * every state carries g and rhs = min over its non-blocked predecessor entries of g(pred) + cost (start: 0);
//   * OPEN = the inconsistent states (g != rhs) under (min(g, rhs) + eps h, min(g, rhs), id), lazy deletion; it is
//     REBUILT from the node pool at the start of every launch, so nothing of it has to survive in LDS or HBM;
//   * successors are re-derived with get_succ (pure) at every expansion; predecessor entries and the blocked log are
//     written by a state's FIRST expansion only (FLAG_BUILT);
//   * map edits: every predecessor entry / log entry is re-evaluated against the current map (edge flag EDGE_BLOCKED,
//     log entries turn into entries), rhs of the states whose entries changed is recomputed;
//   * getSubStateSpace(k): Dijkstra on rhs from the k-th state of the last path through the states that were BUILT,
//     written into a second pool set (which also compacts the pools), then the two sets swap.
// One workgroup per planner (LPA* is a single-query procedure); the pools are private to the planner and flat (chunk
// tables are the identity), so the voxel search's QView / OPEN structure / expand_unit are used unchanged.
#pragma once
#include "mplx_kernels.h"

namespace mplx {

constexpr uint32_t FLAG_BUILT = 4u;   // get_succ has run for this state: its successors hold predecessor entries for it
constexpr uint32_t FLAG_DIRTY = 8u;   // (map edit in progress) a predecessor entry of this state changed
constexpr uint32_t EDGE_BLOCKED = 0x80000000u;  // EdgeRec::action bit: the primitive is not free in the current map
constexpr uint32_t LOG_CONVERTED = 0x80000000u, LOG_FREE_NOW = 0x40000000u;  // blocked-log entry bits (in .y = action)

struct LpaState {  // what persists of one state space besides the pools (HBM)
  uint32_t n_nodes, n_edges, n_blocked, root_id, goal_id, valid, path_len, pad;
  unsigned long long n_changed;     // result of the last update kernel
  uint32_t path[MAX_TRAJ + 1];      // best_child_: node ids of the last trajectory, start -> goal (NIL: dropped)
};
struct LpaParams {
  LpaState *st;
  uint2 *blocked_log;               // (parent id, action | LOG_* bits) of successors emitted with cost +inf, arrival order
  uint32_t blocked_cap;
  int32_t fresh;                    // PLAN: start a new state space (table cleared by the host)
  // SUBTREE: the space being left
  const char *old_node_pool;
  const unsigned long long *old_table;
  unsigned long long old_table_mask;
  const LpaState *old_st;
  int32_t time_step;
};

template <int BLOCK, int CONTROL>
struct LView : QView<BLOCK, CONTROL> {
  static __device__ __forceinline__ double &rhs(char *r) { return *(double *)(r + rec_hot_bytes(CONTROL) + (key_len_c(CONTROL) + 1) * 8); }
};
static_assert(rec_hot_bytes(CTRL_ACC) + (6 + 2) * 8 <= rec_bytes(CTRL_ACC) && rec_hot_bytes(CTRL_JRK) + (9 + 2) * 8 <= rec_bytes(CTRL_JRK) &&
              rec_hot_bytes(CTRL_VEL) + (3 + 2) * 8 <= rec_bytes(CTRL_VEL) && rec_hot_bytes(CTRL_SNP) + (12 + 2) * 8 <= rec_bytes(CTRL_SNP), "rhs fits behind t");

__device__ __forceinline__ bool f64_same(double a, double b) { return __double_as_longlong(a) == __double_as_longlong(b) || a == b; }
__device__ __forceinline__ double lpa_min(double a, double b) { return a < b ? a : b; }

// is_free(Primitive(state, U[action], dt)) by ONE thread, the same expressions as the sampling loop of expand_unit
// (generic branch); also gives the successor state and its key
template <int CONTROL>
__device__ __forceinline__ bool lpa_prim_free(const SearchParams &P, const double *st, int action, State &tn, int32_t *key) {
  constexpr int NQ = nq_c(CONTROL), ns = key_len_c(CONTROL);
  const double T = P.dt;
  double c[3][6];
#pragma unroll
  for (int ax = 0; ax < 3; ax++)
    prim_build_axis(CONTROL, st[ax], ns > 3 ? st[3 + ax] : 0.0, ns > 6 ? st[6 + ax] : 0.0, ns > 9 ? st[9 + ax] : 0.0, P.U[3 * action + ax], c[ax]);
#pragma unroll
  for (int ax = 0; ax < 3; ax++) {
    tn.p[ax] = pos_at_c<CONTROL>(c[ax], T);
    tn.v[ax] = vel_at_c<CONTROL>(c[ax], T);
    tn.a[ax] = acc_at_c<CONTROL>(c[ax], T);
    tn.j[ax] = jrk_at_c<CONTROL>(c[ax], T);
  }
  state_key_c<CONTROL>(tn, key);
  double max_v = 0.0;
  validate_and_maxv_c<CONTROL>(c, T, P.v_max, P.a_max, P.j_max, &max_v);
  const int n = (int)ceil(max_v * T / P.map.res);
  const double dts = n > 0 ? T / n : 0.0;
  double q[3][NQ];
#pragma unroll
  for (int ax = 0; ax < 3; ax++) pack_q_c<CONTROL>(c[ax], q[ax]);
  for (int i = 0; i <= n; i++) {
    const double t = (double)i * dts;
    int32_t cell[3];
    bool in = true;
#pragma unroll
    for (int ax = 0; ax < 3; ax++) {
      cell[ax] = i == 0 ? float_to_cell(st[ax], P.map.origin[ax], P.map.res) : float_to_cell(pos_at_qc<CONTROL>(q[ax], t), P.map.origin[ax], P.map.res);
      in = in && cell[ax] >= 0 && cell[ax] < P.map.dim[ax];
    }
    if (!in) return false;
    if (P.map.data[(size_t)cell[0] + (size_t)P.map.dim[0] * cell[1] + (size_t)P.map.dim[0] * P.map.dim[1] * cell[2]] > 0) return false;
  }
  return true;
}

// rhs of a state from its predecessor entries (state_space.h updateNode); order-independent (an exact minimum)
template <class V, class QV>
__device__ __forceinline__ double lpa_rhs_of(const QV &Q, const SearchParams &P, char *rec) {
  double rhs = INFINITY;
  for (uint32_t e = V::pred(rec); e != NIL; e = Q.edge(e)->next) {
    const EdgeRec er = *Q.edge(e);
    if (er.action & EDGE_BLOCKED) continue;
    const double v = V::g(Q.node(er.parent)) + P.ucost[er.action];
    if (v < rhs) rhs = v;
  }
  return rhs;
}

// table look-up of `key` for query slot 0; returns the node id or NIL.  table / pool may be another space's.
template <int BLOCK, int CONTROL>
__device__ __forceinline__ uint32_t lpa_find(const unsigned long long *table, unsigned long long mask, const char *pool, const int32_t *key, unsigned long long h64) {
  constexpr int nk = key_len_c(CONTROL);
  const unsigned long long tagq = (h64 >> 48) << 48;
  size_t pos = (size_t)h64 & (size_t)mask;
  for (unsigned long long steps = 0; steps <= mask; steps++) {  // (bounded: a full table ends the look-up with "absent")
    const unsigned long long v = ld_u64(&table[pos]);
    if (v == TBL_EMPTY) return NIL;
    const uint32_t vid = (uint32_t)v;
    if (vid < CLAIM_BASE && (v & 0xFFFFFFFF00000000ull) == tagq) {
      const int32_t *kk = (const int32_t *)(pool + (size_t)vid * rec_bytes(CONTROL) + 24);
      uint32_t kd = 0;
#pragma unroll
      for (int i = 0; i < nk; i++) kd |= (uint32_t)(kk[i] ^ key[i]);
      if (kd == 0u) return vid;
    }
    pos = (pos + 1) & (size_t)mask;
  }
  return NIL;
}

template <int BLOCK>
__device__ __forceinline__ void lpa_smem_init(const SearchParams &P, Smem<BLOCK> &S, int tid) {
  for (int i = tid; i < 2 * NB; i += BLOCK) S.cnt[0][i] = 0;
  for (int i = tid; i < MAX_NODE_CH; i += BLOCK) S.node_tbl[i] = (uint16_t)i;  // private flat pools: identity chunk tables
  for (int i = tid; i < MAX_EDGE_CH; i += BLOCK) S.edge_tbl[i] = (uint16_t)i;
  for (int i = tid; i < MAX_OPEN_CH; i += BLOCK) S.open_tbl[i] = (uint16_t)i;
  if (tid == 0) {
    S.n_near = 0; S.n_nodes = 0; S.n_edges = 0; S.n_log = 0;
    S.reserve = (uint32_t)P.n_u + 1u;
    S.node_chunks = P.node_chunks; S.edge_chunks = P.edge_chunks; S.open_chunks = P.open_chunks;
    S.cur1 = 0; S.cur0 = 0; S.lo1 = 0.0; S.ts_f = INFINITY; S.ts_g = INFINITY; S.ts_id = 0xFFFFFFFFu;
    S.f_base = 0.0;
    S.status = -1;
    for (int i = 0; i < 10; i++) S.cyc[i] = 0;
    S.c_expanded = S.c_closed = S.c_prims = S.c_succ = S.c_succ_finite = S.c_reads = 0;
    S.c_push = S.c_reopen = S.c_refill = S.c_evict = 0;
    S.c_hash = 0;
    S.cur_id = NIL;
    S.flag = 0;
  }
}

// pop the smallest VALID entry (lazy deletion).  valid(entry, record) decides; on success S.cur / S.cur_key / S.cur_id /
// S.cur_g (= the record's g) / S.tmp_d0 (= rhs) / S.tmp_u (= flags) describe the state and R holds the entry (key, log
// index).  SUBTREE mode validates against g alone.
struct LpaScratch {  // (LDS) the popped entry and the expanded state's own re-insertion
  double ek, ekg;    // key of the popped entry
  uint32_t eidx;     // its OPEN-log index
  int32_t over;      // the expansion settled an over-consistent state (its g went DOWN to rhs): see the successor update
  int32_t upush;     // the expanded state goes back into OPEN (under-consistent branch) ...
  double uk, ukg;    // ... with this key
};
template <int BLOCK, int CONTROL, bool SUBTREE>
__device__ __forceinline__ bool lpa_pop(const QView<BLOCK, CONTROL> &Q, int tid, double eps, LpaScratch &R) {
  using V = LView<BLOCK, CONTROL>;
  Smem<BLOCK> &S = Q.S;
  constexpr int nk = key_len_c(CONTROL), ns = key_len_c(CONTROL);
  for (;;) {
    if (S.n_near == 0) {
      __syncthreads();
      if (!refill(Q, tid)) return false;
      if (S.n_near == 0) continue;
    }
    const uint32_t n = S.n_near;
    double bf = INFINITY, bg = INFINITY;
    uint32_t bi = 0xFFFFFFFFu, bp = NIL;
    for (uint32_t i = tid; i < n; i += BLOCK) {
      double f = S.near_f[i], g = S.near_g[i];
      uint32_t id = S.near_id[i];
      if (bp == NIL || entry_less(f, g, id, bf, bg, bi)) { bf = f; bg = g; bi = id; bp = i; }
    }
    if (!wave_min_entry(bp != NIL, bf, bg, bi, bp)) bp = NIL;
    if constexpr (BLOCK > 64) {
      if ((tid & 63) == 0) {
        S.red_f[tid >> 6] = bf; S.red_g[tid >> 6] = bg; S.red_id[tid >> 6] = bi; S.red_pos[tid >> 6] = bp;
      }
      __syncthreads();
      bf = S.red_f[0]; bg = S.red_g[0]; bi = S.red_id[0]; bp = S.red_pos[0];
#pragma unroll
      for (int w = 1; w < BLOCK / 64; w++) {
        uint32_t op = S.red_pos[w];
        if (op != NIL && (bp == NIL || entry_less(S.red_f[w], S.red_g[w], S.red_id[w], bf, bg, bi))) {
          bf = S.red_f[w]; bg = S.red_g[w]; bi = S.red_id[w]; bp = op;
        }
      }
      __syncthreads();
    }
    char *rec = Q.node(bi);
    const double rg = V::g(rec), rr = V::rhs(rec), rh = V::h(rec);
    const uint32_t fl = V::flags(rec);
    double sval = 0.0;
    int32_t kval = 0;
    if (tid <= ns) sval = V::state(rec)[tid];
    if (tid < nk) kval = V::key(rec)[tid];
    const uint32_t eidx = S.near_idx[bp];
    __syncthreads();  // everyone has read near_idx[bp] before thread 0 overwrites the slot
    if (tid == 0) {
      const uint32_t last = n - 1;
      S.near_f[bp] = S.near_f[last]; S.near_g[bp] = S.near_g[last];
      S.near_id[bp] = S.near_id[last]; S.near_idx[bp] = S.near_idx[last];
      S.n_near = last;
    }
    bool ok;
    if constexpr (SUBTREE) {
      ok = f64_same(bf, rg) && !(fl & FLAG_BUILT);
    } else {
      const double m = lpa_min(rg, rr);
      ok = !f64_same(rg, rr) && f64_same(bf, m + eps * rh) && f64_same(bg, m);
    }
    if (ok) {
      if (tid <= ns) S.cur[0][tid < ns ? tid : 12] = sval;
      if (tid >= ns && tid < 12) S.cur[0][tid] = 0.0;
      if (tid < nk) S.cur_key[0][tid] = kval;
      if (tid == 0) {
        S.cur_id = bi;
        S.cur_g = rg;
        S.tmp_d0 = rr;
        S.tmp_u = fl;
        R.ek = bf; R.ekg = bg; R.eidx = eidx;
      }
    }
    __syncthreads();
    if (ok) return true;
  }
}

// Look the successors of the expanded state up (act lanes) and, on the state's FIRST expansion, create the missing ones,
// append one predecessor entry per act lane and log the blocked ones.  `only`: lane that may act (-1: all; the serial
// path when two lanes share a key).  Leaves the child's id in child_id (NIL: not present on a repeated expansion).
template <int BLOCK, int CONTROL>
__device__ __forceinline__ void lpa_link(const QView<BLOCK, CONTROL> &Q, const LpaParams &A, int tid, bool act, bool blk, bool first, const LaneSucc &L,
                                         uint32_t &child_id, uint32_t &n_blocked) {
  using V = LView<BLOCK, CONTROL>;
  const SearchParams &P = Q.P;
  Smem<BLOCK> &S = Q.S;
  constexpr int nk = key_len_c(CONTROL), ns = key_len_c(CONTROL);
  int role = 0;  // 1 found, 2 creator
  uint32_t id = NIL;
  size_t tslot = 0;
  unsigned long long h64 = 0, tagq = 0;
  double hspec = 0.0;
  if (act) {
    h64 = key_hash64(L.key, nk);
    tagq = (h64 >> 48) << 48;
    const size_t mask = (size_t)P.table_mask;
    size_t pos = (size_t)h64 & mask;
    const unsigned long long claim = tagq | (unsigned long long)(CLAIM_BASE + (uint32_t)tid);
    for (;;) {
      unsigned long long v = ld_u64(&P.table[pos]);
      if (v == TBL_EMPTY) {
        if (!first) break;  // not there, and a repeated expansion creates nothing
        unsigned long long old = atomicCAS(&P.table[pos], TBL_EMPTY, claim);
        if (old == TBL_EMPTY) { role = 2; tslot = pos; break; }
        v = old;
      }
      const uint32_t vid = (uint32_t)v;
      if (vid < CLAIM_BASE && (v & 0xFFFFFFFF00000000ull) == tagq) {
        const int32_t *kk = V::key(Q.node(vid));
        uint32_t kd = 0;
#pragma unroll
        for (int i = 0; i < nk; i++) kd |= (uint32_t)(kk[i] ^ L.key[i]);
        if (kd == 0u) { role = 1; id = vid; break; }
      }
      pos = (pos + 1) & mask;
    }
    if (role == 2 && P.eps != 0.0) hspec = get_heur(S.hp, CONTROL, L.tn, L.key, nk);
  }
  uint32_t total;
  const uint32_t sc = block_excl_scan<BLOCK>((role == 2 ? 1u : 0u) | (act && first ? 1u << 10 : 0u) | (blk && first ? 1u << 20 : 0u), S, tid, total);
  const uint32_t n_new = total & 0x3FFu, n_fin = (total >> 10) & 0x3FFu, n_blk = total >> 20;
  const uint32_t base_nodes = S.n_nodes, base_edges = S.n_edges, base_blk = n_blocked;
  if (tid == 0) {
    if ((unsigned long long)base_nodes + n_new > ((unsigned long long)P.node_chunks << NODE_CH_LOG) ||
        (unsigned long long)base_edges + n_fin > ((unsigned long long)P.edge_chunks << EDGE_CH_LOG) || (unsigned long long)base_blk + n_blk > A.blocked_cap)
      S.status = 4;  // MPLX_PLAN_POOL_FULL
  }
  __syncthreads();
  if (S.status >= 0) return;
  if (role == 2) {
    id = base_nodes + (sc & 0x3FFu);
    char *rec = Q.node(id);
    int32_t *kk = V::key(rec);
#pragma unroll
    for (int i = 0; i < nk; i++) kk[i] = L.key[i];
    double *st = V::state(rec);
#pragma unroll
    for (int i = 0; i < ns; i++) st[i] = i < 3 ? L.tn.p[i % 3] : i < 6 ? L.tn.v[i % 3] : i < 9 ? L.tn.a[i % 3] : L.tn.j[i % 3];
    st[ns] = S.cur[0][12] + P.dt;
    V::h(rec) = hspec;
    V::g(rec) = INFINITY;
    V::rhs(rec) = INFINITY;
    V::flags(rec) = 0;
    V::pred(rec) = NIL;
    st_u64(&P.table[tslot], tagq | id);
  }
  if (act && first && id != NIL) {
    const uint32_t eidx = base_edges + ((sc >> 10) & 0x3FFu);
    EdgeRec *e = Q.edge(eidx);
    char *rec = Q.node(id);
    e->parent = S.cur_id;
    e->next = V::pred(rec);
    e->action = (uint32_t)tid;
    V::pred(rec) = eidx;
  }
  if (blk && first) A.blocked_log[base_blk + (sc >> 20)] = make_uint2(S.cur_id, (uint32_t)tid);
  child_id = act ? id : NIL;
  if (tid == 0) {
    S.n_nodes = base_nodes + n_new;
    S.n_edges = base_edges + n_fin;
  }
  n_blocked = base_blk + n_blk;
  __syncthreads();
}

// duplicate successor keys inside one expansion?  (sets S.flag; every thread must call)
template <int BLOCK, int CONTROL>
__device__ __forceinline__ void lpa_dup_check(Smem<BLOCK> &S, int tid, bool act, const LaneSucc &L) {
  constexpr int nk = key_len_c(CONTROL);
  S.dupset[tid] = 0;
  S.dupset[tid + BLOCK] = 0;
  if (tid == 0) S.flag = 0;
  __syncthreads();
  if (act) {
    const unsigned long long hv = key_hash64(L.key, nk) | 1ull;
    uint32_t sl = (uint32_t)(hv >> 7) & (2 * BLOCK - 1);
    for (;;) {
      unsigned long long old = atomicCAS(&S.dupset[sl], 0ull, hv);
      if (old == 0ull) break;
      if (old == hv) { S.flag = 1; break; }
      sl = (sl + 1) & (2 * BLOCK - 1);
    }
  }
  __syncthreads();
}

// ------------------------------------------------------------------ ComputeShortestPath
template <int BLOCK, int CONTROL>
__global__ __launch_bounds__(BLOCK) void lpa_plan_kernel(SearchParams P, LpaParams A) {
  __shared__ Smem<BLOCK> S;
  __shared__ uint32_t s_nblk, s_gid, s_root;
  __shared__ int32_t s_stop;
  __shared__ LpaScratch R;
  using V = LView<BLOCK, CONTROL>;
  const int tid = threadIdx.x;
  const QView<BLOCK, CONTROL> Q{P, S, P.bkt_head};
  constexpr int nk = key_len_c(CONTROL), ns = key_len_c(CONTROL);
  const QueryIn &in = P.queries[0];
  fill_uq<BLOCK, CONTROL>(P, S, tid);
  lpa_smem_init<BLOCK>(P, S, tid);
  __syncthreads();
  const unsigned long long t_begin = wall_clock64();
  if (tid == 0) {
    S.hp.w = P.w; S.hp.v_max = P.v_max; S.hp.heur_ignore_dynamics = P.heur_ignore_dynamics;
    S.hp.goal_control = in.goal_control;
    S.hp.goal = in.goal;
    S.hp.goal_nkey = state_key(in.goal_control, in.goal, S.hp.goal_key);
    int32_t c[3];
    bool free_ = true;
    for (int ax = 0; ax < 3; ax++) {
      c[ax] = float_to_cell(in.start.p[ax], P.map.origin[ax], P.map.res);
      if (c[ax] < 0 || c[ax] >= P.map.dim[ax]) free_ = false;
    }
    if (free_) free_ = P.map.data[(size_t)c[0] + (size_t)P.map.dim[0] * c[1] + (size_t)P.map.dim[0] * P.map.dim[1] * c[2]] == 0;
    double cost0 = INFINITY;
    if (!free_)
      S.status = 2;
    else if (in.start_t >= P.t_max || is_goal_state(in.start, in.goal, in.goal_control, P.tol_pos, P.tol_vel, P.tol_acc)) {
      S.status = 0;
      cost0 = 0.0;
    }
    S.tmp_d0 = cost0;
    s_gid = NIL;
    s_root = 0;
    s_nblk = 0;
    if (S.status < 0) {
      if (A.fresh) {  // the start state: g = inf, rhs = 0
        int32_t key[MAX_KEY];
        state_key_c<CONTROL>(in.start, key);
        char *rec = Q.node(0);
        for (int i = 0; i < nk; i++) V::key(rec)[i] = key[i];
        const double *src = (const double *)&in.start;
        for (int i = 0; i < ns; i++) V::state(rec)[i] = src[i];
        V::state(rec)[ns] = in.start_t;
        V::h(rec) = P.eps == 0.0 ? 0.0 : get_heur(S.hp, CONTROL, in.start, key, nk);
        V::g(rec) = INFINITY;
        V::rhs(rec) = 0.0;
        V::flags(rec) = FLAG_OPENED;
        V::pred(rec) = NIL;
        const unsigned long long h64 = key_hash64(key, nk);
        size_t pos = (size_t)h64 & (size_t)P.table_mask;
        for (;;) {
          unsigned long long old = atomicCAS(&P.table[pos], TBL_EMPTY, ((h64 >> 48) << 48) | 0ull);
          if (old == TBL_EMPTY) break;
          pos = (pos + 1) & (size_t)P.table_mask;
        }
        S.n_nodes = 1;
      } else {
        S.n_nodes = A.st->n_nodes;
        S.n_edges = A.st->n_edges;
        s_nblk = A.st->n_blocked;
        s_root = A.st->root_id;
        const uint32_t g0 = A.st->goal_id;
        if (g0 < S.n_nodes) {  // last plan's goal state, if it still is inside the goal region
          const double *st = V::state(Q.node(g0));
          State sg;
          for (int i = 0; i < 12; i++) ((double *)&sg)[i] = i < ns ? st[i] : 0.0;
          if (st[ns] >= P.t_max || is_goal_state(sg, in.goal, in.goal_control, P.tol_pos, P.tol_vel, P.tol_acc)) s_gid = g0;
        }
      }
    }
  }
  __syncthreads();
  uint32_t n_blocked = s_nblk;
  const uint32_t root = s_root;
  bool searched = false;
  if (S.status < 0) {
    searched = true;
    // ---- OPEN = the inconsistent states, rebuilt from the pool (L1); the same pass finds the goal state to follow (L7):
    //      the settled (consistent, finite g) state of the goal REGION with the smallest (key, g, id)
    const uint32_t n0 = S.n_nodes;
    double gk = INFINITY, gg = INFINITY;
    uint32_t gi = 0xFFFFFFFFu, gpos = NIL;  // (gpos != NIL: this thread holds a candidate)
    for (uint32_t base = 0; base < n0; base += BLOCK) {
      while (S.n_near + (uint32_t)BLOCK > (uint32_t)NC) {
        evict_half(Q, tid);
        __syncthreads();
      }
      const uint32_t i = base + tid;
      double g = 0, r = 0, h = 0;
      bool inc = false;
      if (i < n0) {
        char *rec = Q.node(i);
        g = V::g(rec); r = V::rhs(rec); h = V::h(rec);
        inc = !f64_same(g, r);
        if (!inc && g < INFINITY) {  // settled: inside the goal region?
          const double *st = V::state(rec);
          State sg;
          for (int k = 0; k < 12; k++) ((double *)&sg)[k] = k < ns ? st[k] : 0.0;
          if (st[ns] >= P.t_max || is_goal_state(sg, in.goal, in.goal_control, P.tol_pos, P.tol_vel, P.tol_acc)) {
            const double k = g + P.eps * h;
            if (gpos == NIL || entry_less(k, g, i, gk, gg, gi)) { gk = k; gg = g; gi = i; gpos = 0u; }
          }
        }
      }
      uint32_t tot;
      const uint32_t sc = block_excl_scan<BLOCK>(inc ? 1u : 0u, S, tid, tot);
      const uint32_t base_log = S.n_log;
      if (tid == 0 && (unsigned long long)base_log + tot > ((unsigned long long)P.open_chunks << OPEN_CH_LOG)) S.status = 4;
      __syncthreads();
      if (S.status >= 0) break;
      if (inc) {
        const double m = lpa_min(g, r);
        open_push(Q, base_log + sc, m + P.eps * h, m, i);
      }
      __syncthreads();
      if (tid == 0) {
        S.n_log = base_log + tot;
        S.c_push += tot;
      }
      __syncthreads();
    }
    {  // L7: the workgroup's best settled goal-region state, if any, replaces the last plan's goal state
      const bool have = wave_min_entry(gpos != NIL, gk, gg, gi, gpos);
      if ((tid & 63) == 0) { S.red_f[tid >> 6] = gk; S.red_g[tid >> 6] = gg; S.red_id[tid >> 6] = gi; S.red_pos[tid >> 6] = have ? 0u : NIL; }
      __syncthreads();
      if (tid == 0) {
        double bk = INFINITY, bg = INFINITY;
        uint32_t bi = NIL;
        for (int w = 0; w < BLOCK / 64; w++)
          if (S.red_pos[w] != NIL && (bi == NIL || entry_less(S.red_f[w], S.red_g[w], S.red_id[w], bk, bg, bi))) { bk = S.red_f[w]; bg = S.red_g[w]; bi = S.red_id[w]; }
        if (bi != NIL) s_gid = bi;
      }
      __syncthreads();
    }
    // ---- main loop
    uint32_t guard_it = 0;
    while (S.status < 0) {
      if ((++guard_it & 63u) == 0u) {  // launch guard (uniform): heartbeat + the host's abort word, every 64th expansion
        if (tid == 0) {
          guard_mark(P, GUARD_BATCH, 0u, S.c_expanded, (unsigned long long)S.n_nodes);
          if (guard_abort(P)) S.status = PLAN_ABORTED;
        }
        __syncthreads();
        if (S.status >= 0) break;
      }
      while (S.n_near + S.reserve > (uint32_t)NC) {
        evict_half(Q, tid);
        __syncthreads();
      }
      const bool popped = lpa_pop<BLOCK, CONTROL, false>(Q, tid, P.eps, R);
      if (tid == 0) {
        // the goal state's key and consistency (the loop condition of LPAstar)
        double kgoal = INFINITY;
        int gcons = 1, gfin = 0;
        if (s_gid != NIL) {
          char *gr = Q.node(s_gid);
          const double gg = V::g(gr), gq = V::rhs(gr);
          gcons = f64_same(gg, gq) ? 1 : 0;
          gfin = gg < INFINITY ? 1 : 0;
          kgoal = lpa_min(gg, gq) + P.eps * V::h(gr);
        }
        s_stop = 0;
        if (!popped) {
          s_stop = 1;
          S.status = (s_gid != NIL && gcons && gfin) ? 0 : 1;
        } else if (!(R.ek < kgoal || !gcons)) {
          // done: the popped entry is still valid -- it returns to OPEN (only the host-visible open set cares)
          const uint32_t pos = S.n_near;
          S.near_f[pos] = R.ek; S.near_g[pos] = R.ekg; S.near_id[pos] = S.cur_id; S.near_idx[pos] = R.eidx;
          S.n_near = pos + 1;
          s_stop = 1;
          S.status = 0;
        }
      }
      __syncthreads();
      if (s_stop) break;
      const uint32_t u = S.cur_id;
      if (tid == 0) {
        S.c_expanded++;
        S.c_hash = S.c_hash * 0x100000001B3ull + (unsigned long long)(u + 1u);
        if (P.rec_ids && S.c_expanded <= P.cap_rec) P.rec_ids[S.c_expanded - 1] = (int32_t)u;
        char *rec = Q.node(u);
        const double g = S.cur_g, r = S.tmp_d0;
        uint32_t fl = S.tmp_u | FLAG_OPENED | FLAG_CLOSED;
        S.tmp_u = fl & FLAG_BUILT;  // first expansion?
        if (g > r) {
          V::g(rec) = r;  // over-consistent: settle
          S.cur_g = r;
          V::flags(rec) = fl | FLAG_BUILT;
          R.upush = 0;  // no push for u
          R.over = 1;
        } else {
          // under-consistent: g = inf, then the state itself is updated (its key changed)
          V::g(rec) = INFINITY;
          S.cur_g = INFINITY;
          R.over = 0;
          double nr = r;
          if (u != root) nr = lpa_rhs_of<V>(Q, P, rec);
          V::rhs(rec) = nr;
          if (nr < INFINITY) {  // inconsistent again: back into OPEN
            fl &= ~FLAG_CLOSED;
            R.upush = 1;
            R.uk = nr + P.eps * V::h(rec);
            R.ukg = nr;
          } else {
            R.upush = 0;
          }
          V::flags(rec) = fl | FLAG_BUILT;
        }
      }
      __syncthreads();
      const bool first = S.tmp_u == 0u;
      LaneSucc L;
      expand_unit<BLOCK, BLOCK, CONTROL>(P, S, tid, true, L);
      const bool act = L.valid && !L.blocked, blk = L.valid && L.blocked;
      {
        uint32_t tot, treads;
        block_excl_scan<BLOCK>((L.valid ? 1u : 0u) | (act ? 1u << 10 : 0u), S, tid, tot);
        block_excl_scan<BLOCK>(L.reads, S, tid, treads);
        if (tid == 0) {
          S.c_prims += (unsigned long long)P.n_u;
          S.c_succ += tot & 0x3FFu;
          S.c_succ_finite += tot >> 10;
          S.c_reads += treads;
        }
      }
      lpa_dup_check<BLOCK, CONTROL>(S, tid, act, L);
      const bool dup = S.flag != 0;
      uint32_t child = NIL;
      if (!dup) {
        lpa_link<BLOCK, CONTROL>(Q, A, tid, act, blk, first, L, child, n_blocked);
      } else {
        for (int i = 0; i < P.n_u && S.status < 0; i++) {
          uint32_t c2 = NIL;
          lpa_link<BLOCK, CONTROL>(Q, A, tid, act && tid == i, blk && tid == i, first, L, c2, n_blocked);
          if (tid == i) child = c2;
        }
      }
      if (S.status >= 0) break;
      // ---- updateNode(child), one lane per DISTINCT child (the first lane that reaches it)
      bool mine = child != NIL;
      if (dup) S.dupset[tid] = mine ? (unsigned long long)child : ~0ull;  // (rare) two lanes on one child: the lower lane updates
      __syncthreads();
      if (dup && mine) {
        for (int j = 0; j < tid; j++)
          if (S.dupset[j] == (unsigned long long)child) { mine = false; break; }
      }
      bool push = false;
      double pk = 0, pkg = 0;
      if (mine) {
        char *rec = Q.node(child);
        const double g = V::g(rec), old_r = V::rhs(rec);
        double nr = old_r;
        // rhs(child) = min over its predecessor entries of g(pred) + cost.  The stored value is that minimum for the g
        // values before this expansion (every change of a g is followed by this update of all successors); when the
        // expansion LOWERED g(u) and one lane reaches the child, the new minimum is min(old, g(u) + cost of this lane's
        // entry) -- the same f64 sum the walk over the list would form, an exact minimum either way -- so the walk
        // (a chain of dependent HBM reads, most of an expansion's time) is only needed when g(u) went up to inf
        if (child != root) nr = (R.over && !dup) ? lpa_min(old_r, S.cur_g + S.ucost_lds[tid]) : lpa_rhs_of<V>(Q, P, rec);
        uint32_t fl = V::flags(rec);
        const bool was_inc = !f64_same(g, old_r);
        if (child == u) {
          // (a primitive that returns to its own state cannot exist: get_succ drops tn == curr)
        }
        V::rhs(rec) = nr;
        if (!f64_same(g, nr)) {
          const bool had_entry = was_inc && f64_same(nr, old_r);
          fl = (fl | FLAG_OPENED) & ~FLAG_CLOSED;
          if (!had_entry) {
            push = true;
            pkg = lpa_min(g, nr);
            pk = pkg + P.eps * V::h(rec);
          }
        } else if ((fl & FLAG_OPENED) && !(fl & FLAG_CLOSED)) {
          fl |= FLAG_CLOSED;
        }
        V::flags(rec) = fl;
      }
      // u itself (under-consistent branch) pushes through lane n_u (a lane without a primitive) when there is one
      const int ulane = P.n_u < BLOCK ? P.n_u : -1;
      bool upush = false;
      if (tid == ulane && R.upush) {
        upush = true;
        push = true;
        pk = R.uk;
        pkg = R.ukg;
      }
      uint32_t tot;
      const uint32_t sp = block_excl_scan<BLOCK>(push ? 1u : 0u, S, tid, tot);
      const uint32_t base_log = S.n_log;
      if (tid == 0 && (unsigned long long)base_log + tot + 1ull > ((unsigned long long)P.open_chunks << OPEN_CH_LOG)) S.status = 4;
      __syncthreads();
      if (S.status >= 0) break;
      if (push) open_push(Q, base_log + sp, pk, pkg, upush ? u : child);
      __syncthreads();
      if (tid == 0) {
        uint32_t extra = 0;
        if (ulane < 0 && R.upush) {  // (lattice as wide as the workgroup) u's own entry
          open_push(Q, base_log + tot, R.uk, R.ukg, u);
          extra = 1;
        }
        S.n_log = base_log + tot + extra;
        S.c_push += tot + extra;
        State s;
        for (int i = 0; i < 12; i++) ((double *)&s)[i] = S.cur[0][i];
        // (L7: a state popped under-consistent has g = inf now: no goal state to follow)
        if (S.cur_g < INFINITY && (S.cur[0][12] >= P.t_max || is_goal_state(s, in.goal, in.goal_control, P.tol_pos, P.tol_vel, P.tol_acc))) s_gid = u;
        if (P.max_expand > 0 && S.c_expanded >= (unsigned long long)P.max_expand) S.status = 3;
        // (safety net: LPA* expands a state at most twice per call; never spin on the device)
        else if (S.c_expanded > 8ull * ((unsigned long long)P.node_chunks << NODE_CH_LOG) + 1024ull) S.status = 5;
      }
      __syncthreads();
    }
    clear_buckets(Q, tid);
  }
  __syncthreads();
  // closed set size (states that are opened and consistent)
  if (searched) {
    uint32_t cnt = 0;
    for (uint32_t i = tid; i < S.n_nodes; i += BLOCK) cnt += (V::flags(Q.node(i)) & FLAG_CLOSED) ? 1u : 0u;
    uint32_t tot;
    block_excl_scan<BLOCK>(cnt, S, tid, tot);
    if (tid == 0) S.c_closed = tot;
  }
  __syncthreads();
  if (tid == 0) {
    QueryOut &o = P.out[0];
    int32_t *tn = P.traj_nodes, *ta = P.traj_actions;
    double *ts = P.traj_states;
    int status = S.status;
    double cost = INFINITY;
    int len = 0;
    const uint32_t goal_id = s_gid;
    if (status == 0 && !searched) {
      cost = S.tmp_d0;
    } else if (status == 0) {
      uint32_t node = goal_id;
      tn[0] = (int32_t)node;
      bool ok = true, too_long = false;
      while (node != root) {
        uint32_t best = NIL;
        double min_rhs = INFINITY, min_g = INFINITY;
        for (uint32_t e = V::pred(Q.node(node)); e != NIL; e = Q.edge(e)->next) {
          const EdgeRec er = *Q.edge(e);
          if (er.action & EDGE_BLOCKED) continue;
          const double gp = V::g(Q.node(er.parent));
          const double rhs = gp + P.ucost[er.action];
          if (rhs < min_rhs || (rhs == min_rhs && gp >= min_g)) { min_rhs = rhs; min_g = gp; best = e; }
        }
        if (best == NIL || !(min_rhs < INFINITY)) { ok = false; break; }
        if (len >= MAX_TRAJ) { too_long = true; break; }
        ta[len] = (int32_t)Q.edge(best)->action;
        node = Q.edge(best)->parent;
        len++;
        tn[len] = (int32_t)node;
      }
      if (too_long) {
        cost = V::g(Q.node(goal_id));
        status = 6;
        len = 0;
      } else if (ok) {
        cost = V::g(Q.node(goal_id));
        for (int i = 0; i <= len; i++) {
          const double *st = V::state(Q.node((uint32_t)tn[i]));
          for (int k = 0; k < 12; k++) ts[i * 13 + k] = k < ns ? st[k] : 0.0;
          ts[i * 13 + 12] = st[ns];
        }
      } else {
        status = 1;
        len = 0;
      }
    }
    if (searched) {  // the state space stays valid whatever the outcome
      LpaState *st = A.st;
      st->n_nodes = S.n_nodes; st->n_edges = S.n_edges; st->n_blocked = n_blocked;
      st->root_id = root; st->goal_id = goal_id; st->valid = 1;
      if (status == 0) {
        st->path_len = (uint32_t)len;
        for (int i = 0; i <= len; i++) st->path[i] = (uint32_t)tn[len - i];  // start -> goal
      }
    }
    o.status = status;
    o.traj_len = len;
    o.cost = cost;
    o.n_expanded = S.c_expanded; o.n_closed = S.c_closed; o.n_nodes = S.n_nodes; o.n_edges = S.n_edges;
    o.n_primitives = S.c_prims; o.n_succ = S.c_succ; o.n_succ_finite = S.c_succ_finite; o.voxel_reads = S.c_reads;
    o.n_push = S.c_push; o.n_reopen = 0; o.n_refill = S.c_refill; o.n_evict = S.c_evict;
    o.expand_hash = S.c_hash;
    o.n_recorded = (uint32_t)(S.c_expanded < P.cap_rec ? S.c_expanded : P.cap_rec);
    o.slot = 0;
    o.spec[0] = o.spec[1] = o.spec[2] = o.spec[3] = 0;
    o.t_begin = t_begin;
    o.t_end = wall_clock64();
    for (int i = 0; i < 10; i++) o.cyc[i] = 0;
  }
}

// ------------------------------------------------------------------ map edits: increaseCost / decreaseCost
// mode 0: updateBlockedNodes -- entries whose primitive is no longer free get EDGE_BLOCKED
// mode 1: updateClearedNodes -- blocked entries that are free again lose the flag; logged blocked successors whose
//         primitive is free now become entries (in log order, creating the state when it does not exist yet)
// then rhs (and the open / closed flags) of every state whose entries changed is recomputed.
// Three passes, launched one after the other on the planner's stream (round 4: the one-workgroup version took 56 ms for
// the 350 000 entries of the BASELINE C2 state space -- seven times the repair it prepares):
//   pass 0  every predecessor entry (and, mode 1, every entry of the blocked log) re-sampled against the current map --
//           independent of each other, grid-stride over as many workgroups as the launch has; changed entries flag their
//           state (FLAG_DIRTY, a memory-side atomic) and are counted into st->n_changed (zeroed by the host)
//   pass 1  (mode 1) the freed log entries become predecessor entries IN LOG ORDER (ids of the states they create and
//           the order of the entries are results): one thread
//   pass 2  updateNode of the flagged states, grid-stride again
// The passes are separate launches: what pass 0 writes (entry flags, state flags) is read by other workgroups in pass 2.
template <int BLOCK, int CONTROL>
__global__ __launch_bounds__(BLOCK) void lpa_update_kernel(SearchParams P, LpaParams A, int mode, int pass) {
  __shared__ Smem<BLOCK> S;
  using V = LView<BLOCK, CONTROL>;
  const int tid = threadIdx.x;
  const QView<BLOCK, CONTROL> Q{P, S, P.bkt_head};
  constexpr int nk = key_len_c(CONTROL), ns = key_len_c(CONTROL);
  lpa_smem_init<BLOCK>(P, S, tid);
  __syncthreads();
  LpaState *st = A.st;
  if (tid == 0) {
    S.n_nodes = st->n_nodes;
    S.n_edges = st->n_edges;
    S.hp.w = P.w; S.hp.v_max = P.v_max; S.hp.heur_ignore_dynamics = P.heur_ignore_dynamics;
    S.hp.goal_control = P.queries[0].goal_control;
    S.hp.goal = P.queries[0].goal;
    S.hp.goal_nkey = state_key(S.hp.goal_control, S.hp.goal, S.hp.goal_key);
  }
  __syncthreads();
  const uint32_t n_edges0 = S.n_edges, root = st->root_id;
  const uint32_t n_blocked = st->n_blocked;
  const uint32_t gtid = blockIdx.x * BLOCK + (uint32_t)tid, gstride = gridDim.x * BLOCK;
  if (pass == 0) {
    // ---- predecessor entries
    unsigned long long mine = 0;
    for (uint32_t e = gtid; e < n_edges0; e += gstride) {
      EdgeRec *er = Q.edge(e);
      const uint32_t a = er->action;
      const bool was_blocked = (a & EDGE_BLOCKED) != 0;
      if (was_blocked != (mode == 1)) continue;
      State tn;
      int32_t key[MAX_KEY];
      const bool free_ = lpa_prim_free<CONTROL>(P, V::state(Q.node(er->parent)), (int)(a & ~EDGE_BLOCKED), tn, key);
      if (free_ == (mode == 1)) {
        er->action = mode == 1 ? (a & ~EDGE_BLOCKED) : (a | EDGE_BLOCKED);
        const uint32_t child = lpa_find<BLOCK, CONTROL>(P.table, P.table_mask, P.node_pool, key, key_hash64(key, nk));
        if (child != NIL) atomicOr(&V::flags(Q.node(child)), FLAG_DIRTY);
        mine++;
      }
    }
    if (mine) atomicAdd(&st->n_changed, mine);
    // ---- the blocked log (mode 1): evaluated here, converted in log order by pass 1
    if (mode == 1) {
      for (uint32_t b = gtid; b < n_blocked; b += gstride) {
        uint2 le = A.blocked_log[b];
        if (le.y & LOG_CONVERTED) continue;
        State tn;
        int32_t key[MAX_KEY];
        if (lpa_prim_free<CONTROL>(P, V::state(Q.node(le.x)), (int)(le.y & 0xFFFFu), tn, key)) A.blocked_log[b].y = le.y | LOG_FREE_NOW;
      }
    }
    return;
  }
  if (pass == 1) {
    // the freed entries are few among a log of hundreds of thousands: the workgroup scans the log BLOCK entries at a time
    // (one coalesced load each, a ballot), thread 0 converts the marked ones in log order (the serial walk of the whole log
    // by one thread took 37 ms on the C2 state space)
    __shared__ unsigned long long s_mark[BLOCK / 64];
    __shared__ unsigned long long s_conv;
    if (blockIdx.x != 0) return;
    if (tid == 0) s_conv = 0;
    for (uint32_t base = 0; base < n_blocked; base += BLOCK) {
      const uint32_t bi = base + (uint32_t)tid;
      bool want = false;
      if (bi < n_blocked) {
        const uint2 lv = A.blocked_log[bi];
        want = (lv.y & LOG_FREE_NOW) && !(lv.y & LOG_CONVERTED);
      }
      const unsigned long long mk = __ballot(want);
      if ((tid & 63) == 0) s_mark[tid >> 6] = mk;
      __syncthreads();
      if (tid == 0) {
       unsigned long long converted = 0;
       for (int wv = 0; wv < BLOCK / 64 && S.status < 0; wv++)
        for (unsigned long long rest = s_mark[wv]; rest != 0ull && S.status < 0; rest &= rest - 1ull) {
        const uint32_t b = base + (uint32_t)wv * 64u + (uint32_t)(__ffsll((long long)rest) - 1);
        uint2 le = A.blocked_log[b];
        const int action = (int)(le.y & 0xFFFFu);
        State tn;
        int32_t key[MAX_KEY];
        const double *pst = V::state(Q.node(le.x));
        lpa_prim_free<CONTROL>(P, pst, action, tn, key);
        const unsigned long long h64 = key_hash64(key, nk);
        uint32_t id = lpa_find<BLOCK, CONTROL>(P.table, P.table_mask, P.node_pool, key, h64);
        if (id == NIL) {
          if ((unsigned long long)S.n_nodes + 1ull > ((unsigned long long)P.node_chunks << NODE_CH_LOG)) { S.status = 4; break; }
          id = S.n_nodes++;
          char *rec = Q.node(id);
          for (int i = 0; i < nk; i++) V::key(rec)[i] = key[i];
          double *sv = V::state(rec);
          for (int i = 0; i < ns; i++) sv[i] = i < 3 ? tn.p[i % 3] : i < 6 ? tn.v[i % 3] : i < 9 ? tn.a[i % 3] : tn.j[i % 3];
          sv[ns] = pst[ns] + P.dt;
          V::h(rec) = P.eps == 0.0 ? 0.0 : get_heur(S.hp, CONTROL, tn, key, nk);
          V::g(rec) = INFINITY;
          V::rhs(rec) = INFINITY;
          V::flags(rec) = 0;
          V::pred(rec) = NIL;
          size_t pos = (size_t)h64 & (size_t)P.table_mask;
          for (;;) {
            unsigned long long old = atomicCAS(&P.table[pos], TBL_EMPTY, ((h64 >> 48) << 48) | (unsigned long long)id);
            if (old == TBL_EMPTY) break;
            pos = (pos + 1) & (size_t)P.table_mask;
          }
        }
        if ((unsigned long long)S.n_edges + 1ull > ((unsigned long long)P.edge_chunks << EDGE_CH_LOG)) { S.status = 4; break; }
        const uint32_t eidx = S.n_edges++;
        EdgeRec *e = Q.edge(eidx);
        char *rec = Q.node(id);
        e->parent = le.x;
        e->next = V::pred(rec);
        e->action = (uint32_t)action;
        V::pred(rec) = eidx;
        V::flags(rec) |= FLAG_DIRTY;
        A.blocked_log[b].y = (le.y & ~LOG_FREE_NOW) | LOG_CONVERTED;
        converted++;
        }
       s_conv += converted;
      }
      __syncthreads();
      if (S.status >= 0) break;  // (uniform: pool full)
    }
    if (tid == 0) {
      st->n_nodes = S.n_nodes;
      st->n_edges = S.n_edges;
      st->n_changed = S.status == 4 ? ~0ull : st->n_changed + s_conv;  // (~0: pool full -- the host drops the space)
    }
    return;
  }
  // ---- pass 2: updateNode of the states whose entries changed (no OPEN here: the next plan rebuilds it)
  const uint32_t n_nodes = S.n_nodes;
  for (uint32_t i = gtid; i < n_nodes; i += gstride) {
    char *rec = Q.node(i);
    uint32_t fl = V::flags(rec);
    if (!(fl & FLAG_DIRTY)) continue;
    fl &= ~FLAG_DIRTY;
    const double g = V::g(rec);
    double nr = V::rhs(rec);
    if (i != root) nr = lpa_rhs_of<V>(Q, P, rec);
    V::rhs(rec) = nr;
    if (!f64_same(g, nr))
      fl = (fl | FLAG_OPENED) & ~FLAG_CLOSED;
    else if ((fl & FLAG_OPENED) && !(fl & FLAG_CLOSED))
      fl |= FLAG_CLOSED;
    V::flags(rec) = fl;
  }
}

// ------------------------------------------------------------------ getSubStateSpace(time_step)
// Dijkstra on rhs from the time_step-th state of the last path, through the states of the OLD space that had been
// expanded (FLAG_BUILT there), into the NEW space P (table cleared by the host).  Choice L5.
template <int BLOCK, int CONTROL>
__global__ __launch_bounds__(BLOCK) void lpa_subtree_kernel(SearchParams P, LpaParams A) {
  __shared__ Smem<BLOCK> S;
  __shared__ LpaScratch R;
  using V = LView<BLOCK, CONTROL>;
  const int tid = threadIdx.x;
  const QView<BLOCK, CONTROL> Q{P, S, P.bkt_head};
  constexpr int nk = key_len_c(CONTROL), ns = key_len_c(CONTROL);
  fill_uq<BLOCK, CONTROL>(P, S, tid);
  lpa_smem_init<BLOCK>(P, S, tid);
  __syncthreads();
  const LpaState *ost = A.old_st;
  auto old_rec = [&](uint32_t id) { return (char *)A.old_node_pool + (size_t)id * rec_bytes(CONTROL); };
  if (tid == 0) {
    S.hp.w = P.w; S.hp.v_max = P.v_max; S.hp.heur_ignore_dynamics = P.heur_ignore_dynamics;
    S.hp.goal_control = P.queries[0].goal_control;
    S.hp.goal = P.queries[0].goal;
    S.hp.goal_nkey = state_key(S.hp.goal_control, S.hp.goal, S.hp.goal_key);
    // the new root: a copy of the old state, rhs = 0; g = 0 when it had been expanded (it is then traversed below)
    const uint32_t oid = ost->path[A.time_step];
    char *orc = old_rec(oid), *rec = Q.node(0);
    const bool built = (V::flags(orc) & FLAG_BUILT) != 0;
    for (int i = 0; i < nk; i++) V::key(rec)[i] = V::key(orc)[i];
    for (int i = 0; i <= ns; i++) V::state(rec)[i] = V::state(orc)[i];
    V::h(rec) = V::h(orc);
    V::rhs(rec) = 0.0;
    V::g(rec) = built ? 0.0 : INFINITY;
    V::flags(rec) = FLAG_OPENED;
    V::pred(rec) = NIL;
    const unsigned long long h64 = key_hash64(V::key(rec), nk);
    size_t pos = (size_t)h64 & (size_t)P.table_mask;
    for (;;) {
      unsigned long long old = atomicCAS(&P.table[pos], TBL_EMPTY, ((h64 >> 48) << 48) | 0ull);
      if (old == TBL_EMPTY) break;
      pos = (pos + 1) & (size_t)P.table_mask;
    }
    S.n_nodes = 1;
    if (built) {
      open_push(Q, 0u, 0.0, 0.0, 0u);
      S.n_log = 1;
    }
  }
  __syncthreads();
  uint32_t n_blocked = 0;
  uint32_t guard_it = 0;
  while (S.status < 0) {
    if ((++guard_it & 63u) == 0u) {  // launch guard (uniform): heartbeat + the host's abort word
      if (tid == 0) {
        guard_mark(P, GUARD_BATCH, 0u, (unsigned long long)guard_it, (unsigned long long)S.n_nodes);
        if (guard_abort(P)) S.status = PLAN_ABORTED;
      }
      __syncthreads();
      if (S.status >= 0) break;
    }
    while (S.n_near + S.reserve > (uint32_t)NC) {
      evict_half(Q, tid);
      __syncthreads();
    }
    if (!lpa_pop<BLOCK, CONTROL, true>(Q, tid, 0.0, R)) break;
    const uint32_t u = S.cur_id;
    const double gu = S.cur_g;
    LaneSucc L;
    expand_unit<BLOCK, BLOCK, CONTROL>(P, S, tid, true, L);
    const bool act = L.valid && !L.blocked, blk = L.valid && L.blocked;
    lpa_dup_check<BLOCK, CONTROL>(S, tid, act, L);
    const bool dup = S.flag != 0;
    // relax one lane's successor: rhs, and g + an OPEN entry when the state had been expanded in the old space
    auto relax = [&](uint32_t child, bool go, bool &push, double &pk) {
      push = false;
      if (!go || child == NIL) return;
      char *rec = Q.node(child);
      const double tentative = gu + S.ucost_lds[tid];
      if (tentative < V::rhs(rec)) {
        V::rhs(rec) = tentative;
        uint32_t fl = V::flags(rec) | FLAG_OPENED;
        const uint32_t oid = lpa_find<BLOCK, CONTROL>(A.old_table, A.old_table_mask, A.old_node_pool, L.key, key_hash64(L.key, nk));
        const bool obuilt = oid != NIL && (V::flags(old_rec(oid)) & FLAG_BUILT);
        if (obuilt && !(fl & FLAG_BUILT)) {
          V::g(rec) = tentative;
          push = true;
          pk = tentative;
        }
        V::flags(rec) = fl;
      }
    };
    auto push_all = [&](bool push, double pk, uint32_t child) {
      uint32_t tot;
      const uint32_t sp = block_excl_scan<BLOCK>(push ? 1u : 0u, S, tid, tot);
      const uint32_t base_log = S.n_log;
      if (tid == 0 && (unsigned long long)base_log + tot > ((unsigned long long)P.open_chunks << OPEN_CH_LOG)) S.status = 4;
      __syncthreads();
      if (S.status >= 0) return;
      if (push) open_push(Q, base_log + sp, pk, pk, child);
      __syncthreads();
      if (tid == 0) S.n_log = base_log + tot;
      __syncthreads();
    };
    if (!dup) {
      uint32_t child = NIL;
      lpa_link<BLOCK, CONTROL>(Q, A, tid, act, blk, true, L, child, n_blocked);
      if (S.status >= 0) break;
      bool push;
      double pk = 0;
      relax(child, act, push, pk);
      push_all(push, pk, child);
    } else {
      for (int i = 0; i < P.n_u && S.status < 0; i++) {  // in action order, one successor at a time
        uint32_t child = NIL;
        lpa_link<BLOCK, CONTROL>(Q, A, tid, act && tid == i, blk && tid == i, true, L, child, n_blocked);
        if (S.status >= 0) break;
        bool push;
        double pk = 0;
        relax(child, act && tid == i, push, pk);
        push_all(push, pk, child);
      }
    }
    if (S.status >= 0) break;
    if (tid == 0) {
      char *rec = Q.node(u);
      V::flags(rec) |= FLAG_BUILT | FLAG_CLOSED | FLAG_OPENED;
      S.c_expanded++;
    }
    __syncthreads();
  }
  clear_buckets(Q, tid);
  __syncthreads();
  if (tid == 0) {
    LpaState *st = A.st;
    st->n_nodes = S.n_nodes; st->n_edges = S.n_edges; st->n_blocked = n_blocked;
    st->root_id = 0; st->valid = S.status == 4 ? 0u : 1u;
    st->n_changed = S.status == 4 ? ~0ull : S.c_expanded;
    st->path_len = ost->path_len;
    // best_child_ and the goal state in the ids of the new space (NIL: dropped)
    for (uint32_t i = 0; i <= ost->path_len; i++) {
      const uint32_t oid = ost->path[i];
      uint32_t nid = NIL;
      if (oid != NIL) {
        const int32_t *kk = V::key(old_rec(oid));
        int32_t key[MAX_KEY];
        for (int k = 0; k < nk; k++) key[k] = kk[k];
        nid = lpa_find<BLOCK, CONTROL>(P.table, P.table_mask, P.node_pool, key, key_hash64(key, nk));
      }
      st->path[i] = nid;
    }
    uint32_t gid = NIL;
    if (ost->goal_id != NIL && ost->goal_id < ost->n_nodes) {
      const int32_t *kk = V::key(old_rec(ost->goal_id));
      int32_t key[MAX_KEY];
      for (int k = 0; k < nk; k++) key[k] = kk[k];
      gid = lpa_find<BLOCK, CONTROL>(P.table, P.table_mask, P.node_pool, key, key_hash64(key, nk));
    }
    st->goal_id = gid;
  }
}

// ------------------------------------------------------------------ import of a finished A* (round 5)
// The FIRST plan of an LPA* planner is an A* from scratch: same pop order ((min(g, rhs) + eps h, min(g, rhs), id) = (f, g, id)
// while nothing is under-consistent), same states in the same order of creation, same predecessor entries.  The one-workgroup
// LPA* kernel runs it at ~11 us per expansion; astar_spec_kernel with its helper workgroups at ~1.2.  So a fresh LPA* plan is
// planned by the speculative kernel on a private lane context and its state space is IMPORTED into the LPA* pools:
//   states     : record copied; closed (= expanded once) -> g = rhs = g, OPENED | CLOSED | BUILT; open -> rhs = g, g = inf, OPENED
//   entries    : copied (the potential bits of the action word are zero: LPA* refuses an auxiliary map)
//   table      : rebuilt with the LPA* hashing (no query bits)
//   blocked log: the A* keeps no record of successors with cost +inf (D7); re-derived here -- get_succ is pure -- for the
//                expanded states in EXPANSION order (the A*'s record of expanded ids), lanes ascending: the order lpa_link writes
//   LpaState, result record, trajectory: from the A*'s own.
// Everything the replanner does afterwards (updates, repairs, getSubStateSpace) runs on the imported space unchanged;
// tests/test_lpa.py compares g, rhs, h, flags, every entry and its blocked bit with the CPU LPA* after every step.
struct LpaImportArgs {
  const char *node_pool, *edge_pool;        // the A*'s chunked pools
  const uint32_t *node_table, *edge_table;  // chunk tables of its (only) query
  const QueryOut *out;                      // its result record
  const int32_t *rec_ids;                   // expanded node ids in order (out->n_recorded of them)
  const int32_t *traj_nodes, *traj_actions; // goal -> start
  const double *traj_states;
  uint32_t *blk_off;                        // scratch [n_expanded + 1]: blocked successors per expansion, then exclusive offsets
  unsigned long long *blk_mask;             // scratch [2 x n_expanded]: their lanes
  int32_t *dst_rec;                         // the LPA* handle's own expansion record (or null)
  uint32_t dst_cap_rec;
  // mode 1: the imported search is the Dijkstra of getSubStateSpace (FILTER build: eps = 0, no goal, only states that were BUILT in
  // the old space expanded; LpaParams::old_* = the space being left).  A state keeps the heuristic it had in the old space; a
  // state the old space did not hold gets its own (hp: the planner's goal).  No result record, no trajectory: best_child_ and the
  // goal state are re-found in the new space by key.
  // mode 2 (round 6, L5b): the imported search is a fresh A* from the k-th state of the last trajectory -- getSubStateSpace by planning
  // afresh; imported like mode 0, except that the handle's stored trajectory / result stay and best_child_ is re-found by key.
  int32_t mode;
  HeurParams hp;
};

template <int CONTROL>
__global__ __launch_bounds__(256) void lpa_import_copy_kernel(SearchParams P, LpaParams A, LpaImportArgs I) {
  constexpr int nk = key_len_c(CONTROL), ns = key_len_c(CONTROL), rb = rec_bytes(CONTROL);
  using V = LView<256, CONTROL>;
  const QueryOut o = *I.out;
  const size_t n = (size_t)o.n_nodes, ne = (size_t)o.n_edges;
  const size_t gt = (size_t)blockIdx.x * blockDim.x + threadIdx.x, gs = (size_t)gridDim.x * blockDim.x;
  for (size_t i = gt; i < n; i += gs) {
    const char *src = I.node_pool + (((size_t)I.node_table[i >> NODE_CH_LOG] << NODE_CH_LOG) + (i & (((size_t)1 << NODE_CH_LOG) - 1))) * rb;
    char *dst = P.node_pool + i * rb;
    static_assert(rb % 16 == 0, "records are copied in 16-byte words");
#pragma unroll
    for (int k = 0; k < rb / 16; k++) ((uint4 *)dst)[k] = ((const uint4 *)src)[k];
    const double g = V::g(dst);
    const bool closed = (V::flags(dst) & FLAG_CLOSED) != 0;
    V::rhs(dst) = g;
    V::g(dst) = closed ? g : INFINITY;
    V::flags(dst) = FLAG_OPENED | (closed ? (FLAG_CLOSED | FLAG_BUILT) : 0u);
    const unsigned long long h64 = key_hash64(V::key(dst), nk);
    if (I.mode == 1) {  // (the search ran with eps = 0: no heuristic in the record yet)
      int32_t key[MAX_KEY];
#pragma unroll
      for (int k = 0; k < nk; k++) key[k] = V::key(dst)[k];
      const uint32_t oid = lpa_find<256, CONTROL>(A.old_table, A.old_table_mask, A.old_node_pool, key, h64);
      double h = 0.0;
      if (oid != NIL) {
        h = *(const double *)(A.old_node_pool + (size_t)oid * rb + 8);
      } else if (P.eps != 0.0) {
        State st;
        const double *sd = V::state(dst);
#pragma unroll
        for (int k = 0; k < 12; k++) ((double *)&st)[k] = k < ns ? sd[k] : 0.0;
        h = get_heur(I.hp, CONTROL, st, key, nk);
      }
      V::h(dst) = h;
    }
    size_t pos = (size_t)h64 & (size_t)P.table_mask;
    for (unsigned long long steps = 0; steps <= P.table_mask; steps++) {
      if (atomicCAS(&P.table[pos], TBL_EMPTY, ((h64 >> 48) << 48) | (unsigned long long)i) == TBL_EMPTY) break;
      pos = (pos + 1) & (size_t)P.table_mask;
    }
  }
  for (size_t j = gt; j < ne; j += gs) {
    const EdgeRec e = *(const EdgeRec *)(I.edge_pool + (((size_t)I.edge_table[j >> EDGE_CH_LOG] << EDGE_CH_LOG) + (j & (((size_t)1 << EDGE_CH_LOG) - 1))) * EDGE_BYTES);
    EdgeRec *d = (EdgeRec *)(P.edge_pool + j * EDGE_BYTES);
    d->parent = e.parent;
    d->next = e.next;
    d->action = e.action & EDGE_ACTION_MASK;
  }
}

// blocked successors of every expanded state (one workgroup per expansion, grid-stride): lanes as two 64-bit masks + their count
template <int BLOCK, int CONTROL>
__global__ __launch_bounds__(BLOCK) void lpa_import_blocked_kernel(SearchParams P, LpaParams A, LpaImportArgs I) {
  __shared__ Smem<BLOCK> S;
  __shared__ unsigned long long wmask[BLOCK / 64];
  constexpr int nk = key_len_c(CONTROL), ns = key_len_c(CONTROL), rb = rec_bytes(CONTROL);
  using V = LView<BLOCK, CONTROL>;
  const int tid = threadIdx.x;
  fill_uq<BLOCK, CONTROL>(P, S, tid);
  const uint32_t n_exp = I.out->n_recorded;
  for (uint32_t e = blockIdx.x; e < n_exp; e += gridDim.x) {
    char *rec = P.node_pool + (size_t)(uint32_t)I.rec_ids[e] * rb;  // (the imported copy: kernel order on the stream)
    __syncthreads();
    if (tid < 12) S.cur[0][tid] = tid < ns ? V::state(rec)[tid] : 0.0;
    if (tid == 12) S.cur[0][12] = V::state(rec)[ns];
    if (tid < nk) S.cur_key[0][tid] = V::key(rec)[tid];
    __syncthreads();
    LaneSucc L;
    expand_unit<BLOCK, BLOCK, CONTROL>(P, S, tid, true, L);
    const unsigned long long m = __ballot(L.valid && L.blocked);
    if ((tid & 63) == 0) wmask[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
      const unsigned long long m0 = wmask[0], m1 = BLOCK > 64 ? wmask[BLOCK > 64 ? 1 : 0] : 0ull;
      I.blk_mask[2 * (size_t)e] = m0;
      I.blk_mask[2 * (size_t)e + 1] = m1;
      I.blk_off[e] = (uint32_t)(__popcll(m0) + __popcll(m1));
    }
  }
}

// offsets of the log entries (one workgroup: scan with a carry), the entries, then the scalars: LpaState, result, trajectory
template <int CONTROL>
__global__ __launch_bounds__(256) void lpa_import_finish_kernel(SearchParams P, LpaParams A, LpaImportArgs I) {
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const QueryOut o = *I.out;
  const uint32_t n_exp = o.n_recorded;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n_exp; base += 256) {
    const uint32_t e = base + tid;
    const uint32_t v = e < n_exp ? I.blk_off[e] : 0u;
    const uint32_t x = wave_incl_sum<64>(v);
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint32_t pre = carry;
    for (int w = 0; w < wave; w++) pre += wsum[w];
    const uint32_t off = pre + x - v;
    if (e < n_exp && off + v <= A.blocked_cap) {
      const uint32_t id = (uint32_t)I.rec_ids[e];
      uint32_t k = off;
      for (int half = 0; half < 2; half++) {
        unsigned long long m = I.blk_mask[2 * (size_t)e + half];
        while (m) {
          const int b = __ffsll((long long)m) - 1;
          m &= m - 1ull;
          A.blocked_log[k++] = make_uint2(id, (uint32_t)(64 * half + b));
        }
      }
    }
    __syncthreads();
    if (tid == 255) carry = pre + x;
    __syncthreads();
  }
  const uint32_t n_blocked = carry;
  if (I.mode == 1) {  // getSubStateSpace: the scalars of the new space; best_child_ and the goal state re-found by key
    if (tid == 0) {
      constexpr int nk = key_len_c(CONTROL), rb = rec_bytes(CONTROL);
      const LpaState *ost = A.old_st;
      LpaState *st = A.st;
      const bool full = o.status == 4 || n_blocked > A.blocked_cap || o.n_expanded != (unsigned long long)n_exp || o.n_nodes == 0;
      st->n_nodes = (uint32_t)o.n_nodes; st->n_edges = (uint32_t)o.n_edges; st->n_blocked = full ? 0u : n_blocked;
      st->root_id = 0u;
      st->valid = full ? 0u : 1u;
      st->n_changed = full ? ~0ull : o.n_expanded;
      st->path_len = ost->path_len;
      auto new_id = [&](uint32_t oid) {
        if (oid == NIL || oid >= ost->n_nodes) return NIL;
        const int32_t *kk = (const int32_t *)(A.old_node_pool + (size_t)oid * rb + 24);
        int32_t key[MAX_KEY];
        for (int k = 0; k < nk; k++) key[k] = kk[k];
        return lpa_find<256, CONTROL>(P.table, P.table_mask, P.node_pool, key, key_hash64(key, nk));
      };
      for (uint32_t i = 0; i <= ost->path_len; i++) st->path[i] = new_id(ost->path[i]);
      st->goal_id = new_id(ost->goal_id);
    }
    return;
  }
  const bool searched = o.n_nodes > 0;
  const int len = o.status == 0 ? o.traj_len : 0;
  if (I.mode == 2) {  // getSubStateSpace by planning afresh (L5b): the imported A* IS the new space; the handle's stored trajectory and
    // result record are the last plan's and stay; best_child_ is re-found by key, the goal state is the fresh plan's
    if (tid == 0) {
      constexpr int nk = key_len_c(CONTROL), rb = rec_bytes(CONTROL);
      const LpaState *ost = A.old_st;
      LpaState *st = A.st;
      const bool full = !searched || o.status == 4 || n_blocked > A.blocked_cap || o.n_expanded != (unsigned long long)n_exp;
      st->n_nodes = (uint32_t)o.n_nodes; st->n_edges = (uint32_t)o.n_edges; st->n_blocked = full ? 0u : n_blocked;
      st->root_id = 0u;
      st->goal_id = ((o.status == 0 || o.status == 6) && searched) ? (uint32_t)I.traj_nodes[0] : NIL;
      st->valid = full ? 0u : 1u;
      st->n_changed = full ? ~0ull : o.n_expanded;
      st->path_len = ost->path_len;
      for (uint32_t i = 0; i <= ost->path_len; i++) {
        const uint32_t oid = ost->path[i];
        uint32_t nid = NIL;
        if (searched && oid != NIL && oid < ost->n_nodes) {
          const int32_t *kk = (const int32_t *)(A.old_node_pool + (size_t)oid * rb + 24);
          int32_t key[MAX_KEY];
          for (int k = 0; k < nk; k++) key[k] = kk[k];
          nid = lpa_find<256, CONTROL>(P.table, P.table_mask, P.node_pool, key, key_hash64(key, nk));
        }
        st->path[i] = nid;
      }
    }
    return;
  }
  // the trajectory buffers of the LPA* handle (goal -> start, like its own recoverTraj leaves them)
  if (o.status == 0 && searched) {
    for (int i = tid; i <= len; i += 256) P.traj_nodes[i] = I.traj_nodes[i];
    for (int i = tid; i < len; i += 256) P.traj_actions[i] = I.traj_actions[i];
    for (int i = tid; i < (len + 1) * 13; i += 256) P.traj_states[i] = I.traj_states[i];
  }
  if (I.dst_rec)
    for (uint32_t i = tid; i < n_exp && i < I.dst_cap_rec; i += 256) I.dst_rec[i] = I.rec_ids[i];
  if (tid == 0) {
    QueryOut r = o;
    r.slot = 0;
    for (int i = 0; i < 10; i++) r.cyc[i] = 0;
    r.n_recorded = n_exp < I.dst_cap_rec ? n_exp : I.dst_cap_rec;
    LpaState *st = A.st;
    if (searched) {
      const bool full = n_blocked > A.blocked_cap || o.n_expanded != (unsigned long long)n_exp;  // (log or expansion record too small)
      st->n_nodes = (uint32_t)o.n_nodes; st->n_edges = (uint32_t)o.n_edges; st->n_blocked = full ? 0u : n_blocked;
      st->root_id = 0u;
      st->goal_id = ((o.status == 0 || o.status == 6) && searched) ? (uint32_t)I.traj_nodes[0] : NIL;
      st->valid = full ? 0u : 1u;
      if (full) r.status = 4;  // MPLX_PLAN_POOL_FULL
      if (o.status == 0 && !full) {
        st->path_len = (uint32_t)len;
        for (int i = 0; i <= len; i++) st->path[i] = (uint32_t)I.traj_nodes[len - i];  // start -> goal
      }
    }
    P.out[0] = r;
  }
}

}  // namespace mplx
