// mplx_poly_lpa.h -- LPA* on the moving-obstacle environment: PlannerBase::plan with setLPAstar(true), PolyMapPlanner::updateNodes
// and getSubStateSpace as mpl_test_node/src/poly_map_replanner_node.cpp:123-186,231 drives them (round 6).
//
// What the reference does (mpl_external_planner/.../poly_map_planner/poly_map_planner.h:61-93): after the obstacles / the start time
// were changed, updateNodes() walks every state of ss_ptr_->hm_ and every predecessor entry of it, rebuilds the entry's primitive
// (forward_action(pred_coord, pred_action_id)), tests it (isFree(pr, pred_coord.t)) and collects the entries that became blocked
// (cost was finite) or free (cost was +inf) for increaseCost / decreaseCost; plan() then repairs the state space (LPA*).  The search
// and the state space themselves are un-vendored: the algorithm is the one of mplx_lpa.h (Koenig & Likhachev's LPA*, choices L1, L4,
// L6, L7 of DESIGN.md's LPA* section) with two differences that follow the in-tree code:
//   * EVERY successor env_poly_map::get_succ emits is a state with a predecessor entry, also the ones whose primitive is blocked
//     (their entry carries EDGE_BLOCKED = cost +inf): that is what updateNodes walks, and there is no blocked log;
//   * the edge cost depends on the parent state (J(control) + 0.001 J(VEL) + w dt, env_poly_map.h:71-73): it is rebuilt from the
//     parent's record wherever a look-ahead value is formed -- the same expression get_succ evaluates, so the same bits.
// getSubStateSpace(k) is realised by planning afresh from the k-th state of the last trajectory (L5b).
// States are time-keyed (env_poly_map.h:63-64): key = the control kind's integers + round(t / 0.1).
//
// This is the functional version: one 64-lane workgroup; get_succ runs one primitive per lane with the reference's own serial
// collide() loops (obs_point_hits / obs_prim_hits of mplx_poly_dev.h), the state-space bookkeeping of an expansion is done by one
// lane in the order the CPU checker of the tests does it.  A repair is a handful of expansions; a first plan of a few thousand
// expansions takes tens of milliseconds.  (The batched A* of a tick -- astar_poly_kernel -- is the fast path of this environment.)
#pragma once
#include "mplx_lpa.h"
#include "mplx_poly_dev.h"

namespace mplx {

// thread 0's clock per section of an iteration of plpa_plan_kernel -> QueryOut::cyc (mplx_plpa_result_cycles):
// 0 pop, 1 stop test + settling the expanded state, 2 primitives / keys / look-ups / heuristics (lanes), 3 isFree of the primitives
// (poly_collide_all), 4 link (lanes), 5 updateNode of the children: look-ahead values, flags, pushes (lanes), 6 goal test of the expanded state, barrier
#define PLPA_T(k) do { if (tid == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); S.cyc[k] += now_ - t_sec; t_sec = now_; } } while (0)

struct PlpaArgs {
  LpaState *st;
  int32_t fresh;        // start a new state space (the host cleared the table)
  int32_t world;        // index into PolyDev::worlds
  uint32_t *changed;    // (update) edge index | now blocked << 31, appended in no particular order
  uint32_t changed_cap;
  uint32_t *counters;   // (update) [0] entries that became blocked, [1] became free, [2] appended to `changed`, [3] unsupported degree met
  double *edge_cost;    // per predecessor entry: calculate_intrinsic_cost of its primitive, written when the entry is (the parent state never changes)
  // per built state and control input: the successor state and the entry it got when the state was first expanded (NIL: no valid successor)
  uint32_t *succ_child, *succ_entry;
  int32_t trust_entries;  // the entries' blocked bits are those of the world as committed now (a fresh space, or updateNodes ran after the last commit):
                          // a state that is expanded AGAIN reads its successors' outcomes from its entries instead of running get_succ once more
};

// isFree(pr, t) of PolyMapUtil (poly_map_util.h:92-109) for the primitive `cs` that starts at time t_rel (relative to the world's start
// time): the start point against every obstacle, then collide() per obstacle, in the order the obstacles were added (static, linear,
// nonlinear).  1 free, 0 blocked, -1 a hyperplane equation of unsupported degree was met before a hit.
template <bool GEN>
MPLX_HD int plpa_prim_free(const PolyDev &D, const PolyWorld &W, const double cs[2][6], double T, double t_rel) {
  const double x0 = pp_p_auto(cs[0], 0.0), y0 = pp_p_auto(cs[1], 0.0);
  for (int j = 0; j < W.n_obs; j++)
    if (obs_point_hits(D, D.obs[W.obs_off + j], x0, y0, t_rel)) return 0;
  for (int j = 0; j < W.n_obs; j++) {
    const int r = obs_prim_hits<GEN>(D, cs, T, D.obs[W.obs_off + j], t_rel);
    if (r < 0) return -1;
    if (r > 0) return 0;
  }
  return 1;
}

// the primitive of (parent record, action) and its cost
template <int CONTROL, class V>
__device__ __forceinline__ void plpa_edge_prim(const SearchParams &P, char *prec, uint32_t action, double cs[2][6]) {
  constexpr int ns = key_len_c(CONTROL);
  const double *st = V::state(prec);
  const double pos[2] = {st[0], st[1]}, vel[2] = {st[3], st[4]}, acc[2] = {ns > 6 ? st[6] : 0.0, ns > 6 ? st[7] : 0.0};
  const double u[2] = {P.poly.U[2 * action], P.poly.U[2 * action + 1]};
  poly_prim_build(CONTROL, pos, vel, u, cs, acc);
}
// rhs of a state from its non-blocked predecessor entries (an exact minimum: order-independent).  The entry's cost is the value
// get_succ computed when the entry was made -- J(control) + 0.001 J(VEL) + w dt of the primitive from the parent state, which is fixed.
template <int CONTROL, class V, class QV>
__device__ __forceinline__ double plpa_rhs_of(const QV &Q, const double *edge_cost, char *rec) {
  double rhs = INFINITY;
  for (uint32_t e = V::pred(rec); e != NIL;) {
    const EdgeRec er = *Q.edge(e);
    if (!(er.action & EDGE_BLOCKED)) {
      const double v = V::g(Q.node(er.parent)) + edge_cost[e];
      if (v < rhs) rhs = v;
    }
    e = er.next;
  }
  return rhs;
}
// look-up of a time-keyed state in the planner's private table (no query bits: tag = top 16 bits of the key hash)
template <int CONTROL, int NK>
__device__ __forceinline__ uint32_t plpa_find(const SearchParams &P, const int32_t *key, unsigned long long h64, size_t *empty_pos) {
  const unsigned long long tagq = (h64 >> 48) << 48;
  size_t pos = (size_t)h64 & (size_t)P.table_mask;
  for (unsigned long long steps = 0; steps <= P.table_mask; steps++) {
    const unsigned long long v = ld_u64(&P.table[pos]);
    if (v == TBL_EMPTY) {
      if (empty_pos) *empty_pos = pos;
      return NIL;
    }
    const uint32_t vid = (uint32_t)v;
    if ((v & 0xFFFFFFFF00000000ull) == tagq) {
      const int32_t *kk = (const int32_t *)(P.node_pool + (size_t)vid * rec_bytes(CONTROL) + 24);
      uint32_t kd = 0;
      for (int i = 0; i < NK; i++) kd |= (uint32_t)(kk[i] ^ key[i]);
      if (kd == 0u) return vid;
    }
    pos = (pos + 1) & (size_t)P.table_mask;
  }
  if (empty_pos) *empty_pos = (size_t)~0ull;
  return NIL;
}

// ------------------------------------------------------------------ ComputeShortestPath
template <int CONTROL, bool GEN>
__global__ __launch_bounds__(64) void plpa_plan_kernel(SearchParams P, PlpaArgs A) {
  constexpr int BLOCK = 64;
  static_assert(CONTROL == CTRL_ACC || CONTROL == CTRL_JRK, "time-keyed states of the moving-obstacle environment");
  constexpr int ns = key_len_c(CONTROL), NK = ns + 1;
  __shared__ Smem<BLOCK> S;
  __shared__ uint32_t s_gid, s_root;
  __shared__ int32_t s_stop, s_first;
  __shared__ LpaScratch R;
  // the successors of the state being expanded, one per control input
  __shared__ double su_state[POLY_MAX_U][8];  // pos2 vel2 acc2 (+ 2 unused)
  __shared__ int32_t su_key[POLY_MAX_U][MAX_KEY + 1];
  __shared__ int32_t su_valid[POLY_MAX_U], su_blocked[POLY_MAX_U];
  __shared__ double su_cost[POLY_MAX_U];
  __shared__ uint32_t su_id[POLY_MAX_U];             // the successor's state if the table holds it (looked up by its own lane)
  __shared__ unsigned long long su_epos[POLY_MAX_U]; // else the empty slot its probe ended at
  __shared__ uint32_t s_kids[POLY_MAX_U];            // states whose look-ahead value this expansion may have changed, in order
  __shared__ double su_h[POLY_MAX_U];
  __shared__ uint32_t su_pred[POLY_MAX_U], s_fid[POLY_MAX_U], s_feidx[POLY_MAX_U];
  __shared__ int32_t s_nkids;
  __shared__ int32_t s_new[POLY_MAX_U];                   // this input created its successor's state in this expansion ...
  __shared__ unsigned long long s_epos_used[POLY_MAX_U];  // ... in this table slot
  __shared__ unsigned long long plevel[2];                // the time level pprep[] holds (poly_collide_all re-uses it for the next state of that level)
  // isFree(pr, t) of the nine primitives against all obstacles, spread over the lanes below the pair level (poly_collide_all)
  __shared__ double pcs[POLY_MAX_U][2][6];
  __shared__ int32_t phit[POLY_MAX_U], pstart_hit, punsupported, php_max;
  __shared__ PolyPrep pprep[POLY_MAX_OBS];
  __shared__ uint32_t phit_idx[POLY_MAX_U * POLY_MAX_OBS], puns_idx[POLY_MAX_U * POLY_MAX_OBS];
  using V = LView<BLOCK, CONTROL>;
  const int tid = threadIdx.x;
  const QView<BLOCK, CONTROL> Q{P, S, P.bkt_head};
  const QueryIn &in = P.queries[0];
  const PolyDev &D = P.poly;
  const PolyWorld W = D.worlds[A.world];
  lpa_smem_init<BLOCK>(P, S, tid);
  if (tid == 0) { plevel[0] = 0ull; plevel[1] = 0ull; }
  __syncthreads();
  const unsigned long long t_begin = wall_clock64();
  if (tid == 0) {
    S.hp.w = P.w; S.hp.v_max = P.v_max; S.hp.heur_ignore_dynamics = P.heur_ignore_dynamics;
    S.hp.goal_control = in.goal_control;
    S.hp.goal = in.goal;
    S.hp.goal_nkey = state_key(in.goal_control, in.goal, S.hp.goal_key);
    double cost0 = INFINITY;
    if (!poly_inside(W.bbox, 4, in.start.p[0], in.start.p[1]))
      S.status = 2;  // ENV_->is_free(start.pos) failed
    else if (in.start_t >= P.t_max || is_goal_state(in.start, in.goal, in.goal_control, P.tol_pos, P.tol_vel, P.tol_acc)) {
      S.status = 0;
      cost0 = 0.0;
    }
    S.tmp_d0 = cost0;
    s_gid = NIL;
    s_root = 0;
    if (S.status < 0) {
      if (A.fresh) {  // the start state: g = inf, rhs = 0
        int32_t key[MAX_KEY + 1];
        state_key_c<CONTROL>(in.start, key);
        key[ns] = (int32_t)round(in.start_t / 0.1);
        char *rec = Q.node(0);
        for (int i = 0; i < NK; i++) V::key(rec)[i] = key[i];
        const double *src = (const double *)&in.start;
        for (int i = 0; i < ns; i++) V::state(rec)[i] = src[i];
        V::state(rec)[ns] = in.start_t;
        V::h(rec) = P.eps == 0.0 ? 0.0 : get_heur(S.hp, CONTROL, in.start, key, NK);
        V::g(rec) = INFINITY;
        V::rhs(rec) = 0.0;
        V::flags(rec) = FLAG_OPENED;
        V::pred(rec) = NIL;
        const unsigned long long h64 = key_hash64(key, NK);
        st_u64(&P.table[(size_t)h64 & (size_t)P.table_mask], ((h64 >> 48) << 48) | 0ull);  // (the host cleared the table: the home slot is free)
        S.n_nodes = 1;
      } else {
        S.n_nodes = A.st->n_nodes;
        S.n_edges = A.st->n_edges;
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
  const uint32_t root = s_root;
  bool searched = false;
  if (S.status < 0) {
    searched = true;
    // ---- OPEN = the inconsistent states, rebuilt from the pool (L1); the same pass finds the goal state to follow (L7)
    if (tid == 0) {
      double gk = INFINITY, gg = INFINITY;
      uint32_t gi = NIL;
      for (uint32_t i = 0; i < S.n_nodes && S.status < 0; i++) {
        char *rec = Q.node(i);
        const double g = V::g(rec), r = V::rhs(rec), h = V::h(rec);
        if (!f64_same(g, r)) {
          if ((unsigned long long)S.n_log + 1ull > ((unsigned long long)P.open_chunks << OPEN_CH_LOG)) { S.status = 4; break; }
          const double m = lpa_min(g, r);
          // (only entries of the first fine bucket -- f below bucket_width / 1024 -- go to the near set: the rest is linked into far buckets)
          open_push(Q, S.n_log, m + P.eps * h, m, i);
          S.n_log++;
          S.c_push++;
        } else if (g < INFINITY) {
          const double *st = V::state(rec);
          State sg;
          for (int k = 0; k < 12; k++) ((double *)&sg)[k] = k < ns ? st[k] : 0.0;
          if (st[ns] >= P.t_max || is_goal_state(sg, in.goal, in.goal_control, P.tol_pos, P.tol_vel, P.tol_acc)) {
            const double k = g + P.eps * h;
            if (gi == NIL || entry_less(k, g, i, gk, gg, gi)) { gk = k; gg = g; gi = i; }
          }
        }
      }
      if (gi != NIL) s_gid = gi;
    }
    __syncthreads();
    // ---- main loop
    uint32_t guard_it = 0;
    unsigned long long t_sec = __builtin_readcyclecounter();
    while (S.status < 0) {
      if ((++guard_it & 63u) == 0u) {
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
      PLPA_T(0);
      if (tid == 0) {
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
          const uint32_t pos = S.n_near;  // done: the popped entry is still valid -- it returns to OPEN
          S.near_f[pos] = R.ek; S.near_g[pos] = R.ekg; S.near_id[pos] = S.cur_id; S.near_idx[pos] = R.eidx;
          S.n_near = pos + 1;
          s_stop = 1;
          S.status = 0;
        }
      }
      __syncthreads();
      if (s_stop) break;
      const uint32_t u = S.cur_id;
      if (tid == 0) {  // the expanded state itself: settle (over-consistent) or raise to inf and update (under-consistent)
        S.c_expanded++;
        S.c_hash = S.c_hash * 0x100000001B3ull + (unsigned long long)(u + 1u);
        if (P.rec_ids && S.c_expanded <= P.cap_rec) P.rec_ids[S.c_expanded - 1] = (int32_t)u;
        char *rec = Q.node(u);
        const double g = S.cur_g, r = S.tmp_d0;
        uint32_t fl = S.tmp_u | FLAG_OPENED | FLAG_CLOSED;
        s_first = (fl & FLAG_BUILT) ? 0 : 1;
        if (g > r) {
          V::g(rec) = r;
          S.cur_g = r;
        } else {
          V::g(rec) = INFINITY;
          S.cur_g = INFINITY;
          double nr = r;
          if (u != root) nr = plpa_rhs_of<CONTROL, V>(Q, A.edge_cost, rec);
          V::rhs(rec) = nr;
          if (nr < INFINITY) {  // inconsistent again: back into OPEN (its key changed: always a new entry)
            fl &= ~FLAG_CLOSED;
            if ((unsigned long long)S.n_log + 1ull > ((unsigned long long)P.open_chunks << OPEN_CH_LOG)) S.status = 4;
            else {
              open_push(Q, S.n_log, nr + P.eps * V::h(rec), nr, u);
              S.n_log++;
              S.c_push++;
            }
          }
        }
        V::flags(rec) = fl | FLAG_BUILT;
      }
      PLPA_T(1);
      // ---- a state that is expanded AGAIN (it was built before) and whose entries are in step with the world: its successors and
      // whether their primitives are blocked stand in the entries it made when it was first expanded -- nothing of get_succ is run again
      const double T = P.dt, cur_t = S.cur[0][12], t_rel = cur_t - W.start_t;
      const bool again = s_first == 0 && A.trust_entries != 0 && A.succ_child != nullptr;  // (uniform)
      if (again) {
        uint32_t child = NIL;
        bool blocked = false;
        if (tid < P.n_u) {
          const size_t at = (size_t)u * (size_t)P.n_u + (size_t)tid;
          child = A.succ_child[at];
          if (child != NIL) blocked = (Q.edge(A.succ_entry[at])->action & EDGE_BLOCKED) != 0u;
        }
        bool kid = child != NIL && !blocked;
        {  // (two inputs that lead to one state: the child is updated once -- the first of them that is not blocked, as a loop over the inputs finds it)
          const bool kid0 = kid;
          for (int j = 0; j < P.n_u; j++) {  // (uniform)
            const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)child, j);
            const bool kj = __builtin_amdgcn_readlane(kid0 ? 1 : 0, j) != 0;
            if (tid > j && kj && cj == child) kid = false;
          }
        }
        const unsigned long long m_val = __ballot(child != NIL), m_kid = __ballot(kid), m_fin = __ballot(child != NIL && !blocked);
        if (kid) s_kids[__popcll(m_kid & ((1ull << tid) - 1ull))] = child;
        if (tid == 0) {
          S.c_prims += (unsigned long long)P.n_u;
          S.c_succ += (uint32_t)__popcll(m_val);
          S.c_succ_finite += (uint32_t)__popcll(m_fin);
          s_nkids = (int32_t)__popcll(m_kid);
        }
        PLPA_T(2);
      } else {
      // ---- env_poly_map::get_succ(u): lane = control input
      if (tid < P.n_u) {
        const double pos[2] = {S.cur[0][0], S.cur[0][1]}, vel[2] = {S.cur[0][3], S.cur[0][4]}, acc[2] = {S.cur[0][6], S.cur[0][7]};
        const double uu[2] = {D.U[2 * tid], D.U[2 * tid + 1]};
        double c[2][6];
        poly_prim_build(CONTROL, pos, vel, uu, c, acc);
        State tn;
        tn.p[0] = pp_p_auto(c[0], T); tn.p[1] = pp_p_auto(c[1], T); tn.p[2] = 0.0;
        tn.v[0] = pp_v_auto(c[0], T); tn.v[1] = pp_v_auto(c[1], T); tn.v[2] = 0.0;
        for (int k = 0; k < 3; k++) { tn.a[k] = 0.0; tn.j[k] = 0.0; }
        if constexpr (CONTROL == CTRL_JRK) { tn.a[0] = pp_a_auto(c[0], T); tn.a[1] = pp_a_auto(c[1], T); }
        const bool valid = poly_inside(W.bbox, 4, tn.p[0], tn.p[1]) && poly_validate(CONTROL, c, T, P.v_max, P.a_max, P.j_max);
        for (int a = 0; a < 2; a++)
          for (int b = 0; b < 6; b++) pcs[tid][a][b] = c[a][b];
        phit[tid] = 0;
        su_valid[tid] = valid ? 1 : 0;
        su_cost[tid] = poly_intrinsic_cost(CONTROL, c, T, P.w, P.dt);
        su_state[tid][0] = tn.p[0]; su_state[tid][1] = tn.p[1]; su_state[tid][2] = tn.v[0]; su_state[tid][3] = tn.v[1];
        su_state[tid][4] = tn.a[0]; su_state[tid][5] = tn.a[1];
        int32_t key[MAX_KEY + 1];
        state_key_c<CONTROL>(tn, key);
        key[ns] = (int32_t)round((cur_t + P.dt) / 0.1);
        for (int i = 0; i < NK; i++) su_key[tid][i] = key[i];
        // the table look-up of this successor, all lanes at once (the table does not change until the link below)
        size_t epos = 0;
        su_id[tid] = valid ? plpa_find<CONTROL, NK>(P, key, key_hash64(key, NK), &epos) : NIL;
        su_epos[tid] = (unsigned long long)epos;
        su_pred[tid] = su_id[tid] != NIL ? V::pred(Q.node(su_id[tid])) : NIL;  // (head of its predecessor list as it is before this expansion)
        su_h[tid] = (valid && su_id[tid] == NIL && P.eps != 0.0) ? get_heur(S.hp, CONTROL, tn, key, NK) : 0.0;  // (of a state that may have to be created)
      }
      if (tid == 0) { pstart_hit = 0; punsupported = 0; }
      __syncthreads();
      PLPA_T(2);
      // PolyMapUtil::isFree(pr, t) of every valid primitive: the start point against every obstacle, then collide() per obstacle -- the same
      // routine the batched A* of this environment runs (pinned against the compiled reference there), all 64 lanes
      poly_collide_all<BLOCK, PolyNoHook, GEN>(D, W, pcs, su_valid, P.n_u, T, t_rel, pprep, phit_idx, puns_idx, &php_max, phit, &punsupported, &pstart_hit, tid, 1, PolyNoHook(), nullptr,
                                               plevel);
      __syncthreads();
      if (tid < P.n_u) su_blocked[tid] = (su_valid[tid] && (pstart_hit || phit[tid])) ? 1 : 0;
      if (tid == 0 && punsupported) S.status = 5;  // (a hyperplane equation of a degree this build does not solve)
      __syncthreads();
      PLPA_T(3);
      if (S.status >= 0) break;
      // ---- link (first expansion) and the list of children to update.  The lanes do it side by side -- ids of new states and entry
      // numbers from prefix counts in input order, i.e. what a loop over the inputs hands out -- unless two inputs of this expansion meet
      // in one state or in one empty table slot (equal key hashes / equal slots: rare); then one lane does it in the order get_succ emits.
      static_assert(BLOCK == 64, "one wavefront: ballots and lane reads span the workgroup");
      bool lanes_linked = false;
      {
        const bool first = s_first != 0;
        const bool mv = tid < P.n_u && su_valid[tid < POLY_MAX_U ? tid : 0] != 0;
        const int ti = tid < POLY_MAX_U ? tid : 0;
        uint32_t my_id = mv ? su_id[ti] : NIL;
        const unsigned long long my_epos = mv ? su_epos[ti] : 0ull;
        const unsigned long long my_h64 = mv ? key_hash64(su_key[ti], NK) : 0ull;
        bool clash = false;
        for (int j = 0; j < P.n_u; j++) {  // (uniform)
          const unsigned long long hj = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_h64 >> 32), j) << 32) |
                                        (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_h64, j);
          const unsigned long long ej = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_epos >> 32), j) << 32) |
                                        (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_epos, j);
          const bool vj = __builtin_amdgcn_readlane(mv ? 1 : 0, j) != 0;
          const uint32_t idj = (uint32_t)__builtin_amdgcn_readlane((int)my_id, j);
          if (mv && vj && tid > j && (hj == my_h64 || (my_id == NIL && idj == NIL && ej == my_epos))) clash = true;
        }
        const unsigned long long below = (1ull << tid) - 1ull;
        const unsigned long long m_val = __ballot(mv), m_new = __ballot(mv && first && my_id == NIL);
        const uint32_t n_val = (uint32_t)__popcll(m_val), n_new = (uint32_t)__popcll(m_new);
        const bool full = first && (__ballot(mv && my_id == NIL && my_epos == ~0ull) != 0ull ||
                                    (unsigned long long)S.n_nodes + n_new > ((unsigned long long)P.node_chunks << NODE_CH_LOG) ||
                                    (unsigned long long)S.n_edges + n_val > ((unsigned long long)P.edge_chunks << EDGE_CH_LOG));
        if (__ballot(clash) == 0ull && !full) {  // (uniform)
          lanes_linked = true;
          const uint32_t base_n = S.n_nodes, base_e = S.n_edges;
          uint32_t my_eidx = NIL;
          if (first && mv) {
            const bool is_new = my_id == NIL;
            if (is_new) {
              my_id = base_n + (uint32_t)__popcll(m_new & below);
              char *rec = Q.node(my_id);
              for (int k = 0; k < NK; k++) V::key(rec)[k] = su_key[ti][k];
              double *st = V::state(rec);
              st[0] = su_state[ti][0]; st[1] = su_state[ti][1]; st[2] = 0.0;
              st[3] = su_state[ti][2]; st[4] = su_state[ti][3]; st[5] = 0.0;
              if constexpr (ns > 6) { st[6] = su_state[ti][4]; st[7] = su_state[ti][5]; st[8] = 0.0; }
              st[ns] = cur_t + P.dt;
              V::h(rec) = su_h[ti];
              V::g(rec) = INFINITY;
              V::rhs(rec) = INFINITY;
              V::flags(rec) = 0;
              st_u64(&P.table[(size_t)my_epos], ((my_h64 >> 48) << 48) | (unsigned long long)my_id);
            }
            const uint32_t eidx = base_e + (uint32_t)__popcll(m_val & below);
            my_eidx = eidx;
            EdgeRec *e = Q.edge(eidx);
            e->parent = u;
            e->next = is_new ? NIL : su_pred[ti];
            e->action = (uint32_t)tid | (su_blocked[ti] ? EDGE_BLOCKED : 0u);
            A.edge_cost[eidx] = su_cost[ti];
            V::pred(Q.node(my_id)) = eidx;
          }
          if (first && A.succ_child && tid < P.n_u) {  // (what a later expansion of u reads instead of running get_succ again)
            const size_t at = (size_t)u * (size_t)P.n_u + (size_t)tid;
            A.succ_child[at] = mv ? my_id : NIL;
            A.succ_entry[at] = my_eidx;
          }
          const bool kid = mv && my_id != NIL && !su_blocked[ti];
          const unsigned long long m_kid = __ballot(kid);
          if (kid) s_kids[__popcll(m_kid & below)] = my_id;
          if (tid == 0) {
            if (first) { S.n_nodes = base_n + n_new; S.n_edges = base_e + n_val; }
            S.c_prims += (unsigned long long)P.n_u;
            S.c_succ += n_val;
            S.c_succ_finite += (uint32_t)__popcll(m_kid);
            s_nkids = (int32_t)__popcll(m_kid);
          }
        }
      }
      if (!lanes_linked && tid == 0) {
        const bool first = s_first != 0;
        uint32_t *kids = s_kids;
        int nk_ = 0;
        uint32_t n_valid = 0, n_fin = 0;
        for (int i = 0; i < P.n_u; i++) s_new[i] = 0;
        if (first && A.succ_child)
          for (int i = 0; i < P.n_u; i++) A.succ_child[(size_t)u * (size_t)P.n_u + (size_t)i] = NIL;
        for (int i = 0; i < P.n_u && S.status < 0; i++) {
          if (!su_valid[i]) continue;
          n_valid++;
          const unsigned long long h64 = key_hash64(su_key[i], NK);
          size_t epos = (size_t)su_epos[i];
          uint32_t id = su_id[i];
          if (id == NIL) {
            // the lane's look-up saw the table as it was before this expansion: an earlier input may have created this very state since
            // (same key: compared here), or taken the empty slot the lane's probe ended at (then, and only then, the probe is repeated)
            bool again = false;
            for (int j = 0; j < i; j++) {
              if (!s_new[j]) continue;
              bool same = true;
              for (int k = 0; k < NK; k++) same = same && su_key[j][k] == su_key[i][k];
              if (same) id = s_fid[j];
              else if (s_epos_used[j] == (unsigned long long)epos) again = true;
            }
            if (id == NIL && again) id = plpa_find<CONTROL, NK>(P, su_key[i], h64, &epos);
          }
          if (first) {
            if (id == NIL) {
              if (epos == (size_t)~0ull || (unsigned long long)S.n_nodes + 1ull > ((unsigned long long)P.node_chunks << NODE_CH_LOG)) { S.status = 4; break; }
              id = S.n_nodes++;
              char *rec = Q.node(id);
              for (int k = 0; k < NK; k++) V::key(rec)[k] = su_key[i][k];
              // (the record's state: pos3 vel3 [acc3], then t -- written field by field: a State object addressed through a pointer would live
              //  in scratch memory, and every read of it would wait for all the stores in flight)
              double *st = V::state(rec);
              st[0] = su_state[i][0]; st[1] = su_state[i][1]; st[2] = 0.0;
              st[3] = su_state[i][2]; st[4] = su_state[i][3]; st[5] = 0.0;
              if constexpr (ns > 6) { st[6] = su_state[i][4]; st[7] = su_state[i][5]; st[8] = 0.0; }
              st[ns] = cur_t + P.dt;
              V::h(rec) = su_h[i];
              V::g(rec) = INFINITY;
              V::rhs(rec) = INFINITY;
              V::flags(rec) = 0;
              V::pred(rec) = NIL;
              st_u64(&P.table[epos], ((h64 >> 48) << 48) | (unsigned long long)id);
              s_new[i] = 1;
              s_epos_used[i] = (unsigned long long)epos;
            }
            if ((unsigned long long)S.n_edges + 1ull > ((unsigned long long)P.edge_chunks << EDGE_CH_LOG)) { S.status = 4; break; }
            const uint32_t eidx = S.n_edges++;
            EdgeRec *e = Q.edge(eidx);
            char *rec = Q.node(id);
            // the list's head: the entry an earlier input of this expansion gave the same state, else what the lane read (NIL for a new state)
            uint32_t head = id == su_id[i] ? su_pred[i] : NIL;
            for (int j = 0; j < i; j++)
              if (su_valid[j] && s_fid[j] == id) head = s_feidx[j];
            s_fid[i] = id; s_feidx[i] = eidx;
            e->parent = u;
            e->next = head;
            e->action = (uint32_t)i | (su_blocked[i] ? EDGE_BLOCKED : 0u);
            A.edge_cost[eidx] = su_cost[i];
            V::pred(rec) = eidx;
            if (A.succ_child) {
              A.succ_child[(size_t)u * (size_t)P.n_u + (size_t)i] = id;
              A.succ_entry[(size_t)u * (size_t)P.n_u + (size_t)i] = eidx;
            }
          } else if (id == NIL) {
            continue;
          }
          if (su_blocked[i]) continue;  // (a blocked entry cannot lower the successor's look-ahead value)
          n_fin++;
          bool dup = false;
          for (int j = 0; j < nk_; j++) dup = dup || kids[j] == id;
          if (!dup) kids[nk_++] = id;
        }
        S.c_prims += (unsigned long long)P.n_u;
        S.c_succ += n_valid;
        S.c_succ_finite += n_fin;
        s_nkids = S.status < 0 ? nk_ : 0;
      }
      }
      __syncthreads();
      PLPA_T(4);
      // updateNode of every child, one lane each (entries and g values are final for this expansion; the children are distinct states):
      // look-ahead value, flags, and -- where its key changed or it has no entry -- a new OPEN entry, whose place in the log is its rank
      // among the pushing children in input order
      {
        const int nk_ = s_nkids;  // (uniform)
        const bool mine = tid < nk_;
        const uint32_t id = mine ? s_kids[tid] : NIL;
        char *rec = mine ? Q.node(id) : nullptr;
        double g = 0.0, nr = 0.0, h = 0.0;
        uint32_t fl = 0u;
        bool need = false;
        if (mine) {
          const double old_r = V::rhs(rec);
          g = V::g(rec); fl = V::flags(rec); h = V::h(rec);
          nr = id != root ? plpa_rhs_of<CONTROL, V>(Q, A.edge_cost, rec) : old_r;
          if (!f64_same(g, nr)) {
            const bool had_entry = (fl & FLAG_OPENED) && !(fl & FLAG_CLOSED) && f64_same(nr, old_r);
            fl = (fl | FLAG_OPENED) & ~FLAG_CLOSED;
            need = !had_entry;
          } else if ((fl & FLAG_OPENED) && !(fl & FLAG_CLOSED)) {
            fl |= FLAG_CLOSED;
          }
        }
        const unsigned long long m_push = __ballot(need);
        const uint32_t n_push = (uint32_t)__popcll(m_push), base_log = S.n_log;
        if ((unsigned long long)base_log + n_push > ((unsigned long long)P.open_chunks << OPEN_CH_LOG)) {  // (uniform)
          if (tid == 0) S.status = 4;
        } else {
          if (mine) {
            V::rhs(rec) = nr;
            V::flags(rec) = fl;
            if (need) {
              const double m = lpa_min(g, nr);
              open_push(Q, base_log + (uint32_t)__popcll(m_push & ((1ull << tid) - 1ull)), m + P.eps * h, m, id);
            }
          }
          if (tid == 0) { S.n_log = base_log + n_push; S.c_push += n_push; }
        }
      }
      PLPA_T(5);
      if (tid == 0) {
        State s;  // (field by field: a State addressed through a pointer would live in scratch memory)
        s.p[0] = S.cur[0][0]; s.p[1] = S.cur[0][1]; s.p[2] = S.cur[0][2]; s.v[0] = S.cur[0][3]; s.v[1] = S.cur[0][4]; s.v[2] = S.cur[0][5];
        s.a[0] = S.cur[0][6]; s.a[1] = S.cur[0][7]; s.a[2] = S.cur[0][8]; s.j[0] = S.cur[0][9]; s.j[1] = S.cur[0][10]; s.j[2] = S.cur[0][11];
        if (S.cur_g < INFINITY && (S.cur[0][12] >= P.t_max || is_goal_state(s, in.goal, in.goal_control, P.tol_pos, P.tol_vel, P.tol_acc))) s_gid = u;
        if (S.status < 0) {
          if (P.max_expand > 0 && S.c_expanded >= (unsigned long long)P.max_expand) S.status = 3;
          else if (S.c_expanded > 8ull * ((unsigned long long)P.node_chunks << NODE_CH_LOG) + 1024ull) S.status = 5;
        }
      }
      __syncthreads();
      PLPA_T(6);
    }
    clear_buckets(Q, tid);
  }
  __syncthreads();
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
      while (node != root) {  // recoverTraj: min g(pred) + cost over the non-blocked entries, ties -> larger g(pred), then the oldest entry
        uint32_t best = NIL;
        double min_rhs = INFINITY, min_g = INFINITY;
        for (uint32_t e = V::pred(Q.node(node)); e != NIL; e = Q.edge(e)->next) {
          const EdgeRec er = *Q.edge(e);
          if (er.action & EDGE_BLOCKED) continue;
          const double gp = V::g(Q.node(er.parent)), rhs = gp + A.edge_cost[e];
          if (rhs < min_rhs || (rhs == min_rhs && gp >= min_g)) { min_rhs = rhs; min_g = gp; best = e; }
        }
        if (best == NIL || !(min_rhs < INFINITY)) { ok = false; break; }
        if (len >= MAX_TRAJ) { too_long = true; break; }
        ta[len] = (int32_t)(Q.edge(best)->action & ~EDGE_BLOCKED);
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
    if (searched) {
      LpaState *st = A.st;
      st->n_nodes = S.n_nodes; st->n_edges = S.n_edges; st->n_blocked = 0;
      st->root_id = root; st->goal_id = goal_id; st->valid = 1;
      if (status == 0) {
        st->path_len = (uint32_t)len;
        for (int i = 0; i <= len; i++) st->path[i] = (uint32_t)tn[len - i];
      }
    }
    o.status = status;
    o.traj_len = len;
    o.cost = cost;
    o.n_expanded = S.c_expanded; o.n_closed = S.c_closed; o.n_nodes = S.n_nodes; o.n_edges = S.n_edges;
    o.n_primitives = S.c_prims; o.n_succ = S.c_succ; o.n_succ_finite = S.c_succ_finite; o.voxel_reads = 0;
    o.n_push = S.c_push; o.n_reopen = 0; o.n_refill = S.c_refill; o.n_evict = S.c_evict;
    o.expand_hash = S.c_hash;
    o.n_recorded = (uint32_t)(S.c_expanded < P.cap_rec ? S.c_expanded : P.cap_rec);
    o.slot = 0;
    o.spec[0] = o.spec[1] = o.spec[2] = o.spec[3] = 0;
    o.t_begin = t_begin;
    o.t_end = wall_clock64();
    for (int i = 0; i < 10; i++) o.cyc[i] = S.cyc[i];
  }
}

// ------------------------------------------------------------------ PolyMapPlanner::updateNodes (poly_map_planner.h:61-93)
// One lane per state: every predecessor entry re-tested against the current obstacles (forward_action + isFree(pr, pred.t)); an
// entry whose outcome changed flips its EDGE_BLOCKED bit (increaseCost / decreaseCost) and is reported; the look-ahead value of a
// state whose entries changed is recomputed.  (g values do not change here, so the states are independent of each other.)
template <int CONTROL, bool GEN>
__global__ __launch_bounds__(64) void plpa_update_kernel(SearchParams P, PlpaArgs A) {
  constexpr int BLOCK = 64;
  constexpr int ns = key_len_c(CONTROL);
  __shared__ Smem<BLOCK> S;
  using V = LView<BLOCK, CONTROL>;
  const int tid = threadIdx.x;
  const QView<BLOCK, CONTROL> Q{P, S, P.bkt_head};
  lpa_smem_init<BLOCK>(P, S, tid);
  __syncthreads();
  const PolyDev &D = P.poly;
  const PolyWorld W = D.worlds[A.world];
  const uint32_t n = A.st->n_nodes, root = A.st->root_id;
  for (uint32_t i = blockIdx.x * BLOCK + tid; i < n; i += gridDim.x * BLOCK) {
    char *rec = Q.node(i);
    bool changed = false;
    for (uint32_t e = V::pred(rec); e != NIL; e = Q.edge(e)->next) {
      EdgeRec *er = Q.edge(e);
      const uint32_t a = er->action;
      char *prec = Q.node(er->parent);
      double cs[2][6];
      plpa_edge_prim<CONTROL, V>(P, prec, a & ~EDGE_BLOCKED, cs);
      const int fr = plpa_prim_free<GEN>(D, W, cs, P.dt, V::state(prec)[ns] - W.start_t);
      if (fr < 0) { atomicAdd(&A.counters[3], 1u); continue; }
      const bool blocked_now = fr == 0, was = (a & EDGE_BLOCKED) != 0u;
      if (blocked_now != was) {
        er->action = blocked_now ? (a | EDGE_BLOCKED) : (a & ~EDGE_BLOCKED);
        changed = true;
        atomicAdd(&A.counters[blocked_now ? 0 : 1], 1u);
        const uint32_t k = atomicAdd(&A.counters[2], 1u);
        if (k < A.changed_cap) A.changed[k] = e | (blocked_now ? 0x80000000u : 0u);
      }
    }
    if (changed && i != root) V::rhs(rec) = plpa_rhs_of<CONTROL, V>(Q, A.edge_cost, rec);
  }
}

}  // namespace mplx
