// mplx_poly_search.h -- astar_poly_kernel: GraphSearch::Astar over the moving-obstacle environment (env_poly_map),
// one workgroup per query, device resident like astar_kernel (mplx_kernels.h) whose OPEN structure, state-space
// pools, hash table, commit and recoverTraj it shares.  What differs from the voxel environment:
//   * get_succ is env_poly_map::get_succ (mplx_poly_dev.h): lane i < n_u builds primitive i, the (primitive,
//     obstacle) pairs are spread over the lanes, one collide() each;
//   * successors carry time: tn.t = curr.t + dt and enable_t (env_poly_map.h:63-64), so the state key has one more
//     integer, round(t / 0.1), and a state revisited at another time is another node;
//   * the edge cost depends on the state (J(control) + 0.001 J(VEL) + w dt, env_poly_map.h:71-73), so it travels
//     with the lane and recoverTraj recomputes it from the parent state;
//   * PlannerBase::plan's start test is ENV_->is_free(start.pos) = inside the bounding box (env_poly_map.h:33).
// States are stored in the 3-D record layout with z = 0 (like the occupancy-map planner).
#pragma once
#include "mplx_kernels.h"

namespace mplx {

// CONTROL: ACC (the multi-robot node) or JRK (poly_map_planner_node.cpp:73-85, use_acc); GEN: hyperplane equations above
// degree two can occur (JRK primitives, obstacle trajectories with cubic or higher segments): solve_any6
template <int BLOCK, int CONTROL = CTRL_ACC, bool GEN = false>
__global__ __launch_bounds__(BLOCK) void astar_poly_kernel(SearchParams P) {
  static_assert(CONTROL == CTRL_ACC || CONTROL == CTRL_JRK, "time-keyed states: the key of an SNP state would need 13 integers");
  constexpr int ns = key_len_c(CONTROL), NK = ns + 1;
  __shared__ Smem<BLOCK> S;
  __shared__ double pcs[POLY_MAX_U][2][6];
  __shared__ int32_t pvalid[POLY_MAX_U], phit[POLY_MAX_U];
  __shared__ int32_t pstart_hit, punsupported, php_max;
  __shared__ PolyPrep pprep[POLY_MAX_OBS];
  __shared__ uint32_t phit_idx[POLY_MAX_U * POLY_MAX_OBS], puns_idx[POLY_MAX_U * POLY_MAX_OBS];
  __shared__ PolyWorldLds wlds;
  __shared__ unsigned long long plevel[2];  // the (query, time level) pprep[] holds
  __shared__ unsigned long long pmask;      // (helpers) the look-ahead record of the state being expanded
  __shared__ double hstate[8];              // (helper) the ring entry being served
  __shared__ int32_t hgo;
  __shared__ double pU[POLY_MAX_U][2];      // the control inputs (D.U is a global load at the head of every expansion otherwise)
  using V = QView<BLOCK, CONTROL>;
  const int tid = threadIdx.x;
  const PolyDev &DG = P.poly;
  const int n_help = DG.n_help, n_lead = P.nq, first_helper = P.help_lead;
  // ---- helper workgroups (blockIdx >= first_helper; the host launches them only when every leader has exactly one query):
  // helper j of leader `slot` serves the states id = 1 + j, 1 + j + n_help, ... of that leader's search as they are
  // published -- primitives, then the collision tests, exactly what the leader would do on popping the state -- and
  // leaves the outcome in help_mask[record].  It never writes anything else, the leader never waits for it.
  // (the helpers may be a launch of their own -- first_helper = 0 -- so that leaders and helpers can differ in width)
  if (n_help > 0 && (int)blockIdx.x >= first_helper) {
    const int hidx = (int)blockIdx.x - first_helper, slot = hidx % n_lead, hj = hidx / n_lead;
    const int q = P.order[slot];
    const PolyWorld WG = DG.worlds[P.poly_world[q]];
    PolyDev D;
    PolyWorld W;
    poly_stage_world<BLOCK>(DG, WG, wlds, tid, D, W);
    if (tid < P.n_u) { pU[tid][0] = D.U[2 * tid]; pU[tid][1] = D.U[2 * tid + 1]; }  // (read after the __syncthreads() that follow)
    if (tid == 0) { plevel[0] = 0ull; plevel[1] = 0ull; punsupported = 0; }
    const unsigned long long ring_mask = (1ull << DG.help_ring_log) - 1ull;
    const double *ring = DG.help_ring + ((size_t)slot << DG.help_ring_log) * 8;
    unsigned long long next = 1ull + (unsigned long long)hj;
    for (;;) {
      if (tid == 0) {
        int go = 0;
        for (int spin = 0; spin < 4000000; spin++) {  // (a bound on the polls, not on time: a leader that has stopped publishing is gone)
          const unsigned long long pv = ld_u64(&DG.help_pub[slot]);
          const unsigned long long n = pv & ~POLY_PUB_DONE;
          if (next < n) {
            if (n - next > ring_mask) next += ((n - ring_mask - next + (unsigned long long)n_help - 1ull) / (unsigned long long)n_help) * (unsigned long long)n_help;  // lapped: skip
            if (next < n) { go = 1; break; }
          }
          if (pv & POLY_PUB_DONE) break;
          if ((spin & 4095) == 4095 && guard_abort(P)) break;  // launch guard: the host has given up on this launch
          __builtin_amdgcn_s_sleep(32);
        }
        hgo = go;
      }
      __syncthreads();
      if (!hgo) break;
      if (tid < 8) hstate[tid] = ld_f64_agent(&ring[(next & ring_mask) * 8 + (unsigned long long)tid]);
      if (tid == 0) { pstart_hit = 0; punsupported = 0; }
      __syncthreads();
      const unsigned long long tag = (unsigned long long)__double_as_longlong(hstate[0]);
      if ((tag >> 32) == (next & 0xFFFFFFFFull)) {  // (uniform) the entry of state `next` (not yet overwritten by a later lap)
        const double T = P.dt, cur_t = hstate[7], t_rel = cur_t - W.start_t;
        if (tid < P.n_u) {
          const double pos[2] = {hstate[1], hstate[2]}, vel[2] = {hstate[3], hstate[4]}, acc[2] = {hstate[5], hstate[6]}, u[2] = {pU[tid][0], pU[tid][1]};
          double c[2][6];
          poly_prim_build(CONTROL, pos, vel, u, c, acc);
          for (int i = 0; i < 2; i++)
            for (int j = 0; j < 6; j++) pcs[tid][i][j] = c[i][j];
          const double ex = pp_p_auto(c[0], T), ey = pp_p_auto(c[1], T);
          pvalid[tid] = (poly_inside(W.bbox, 4, ex, ey) && poly_validate(CONTROL, c, T, P.v_max, P.a_max, P.j_max)) ? 1 : 0;
          phit[tid] = 0;
        }
        __syncthreads();
        poly_collide_all<BLOCK, PolyNoHook, GEN>(D, W, pcs, pvalid, P.n_u, T, t_rel, pprep, phit_idx, puns_idx, &php_max, phit, &punsupported, &pstart_hit, tid, (long long)q + 1,
                                                 PolyNoHook(), nullptr, plevel);
        if (tid == 0) {
          unsigned long long m = POLY_MASK_READY;
          for (int i = 0; i < P.n_u; i++)
            if (phit[i]) m |= 1ull << i;
          if (pstart_hit) m |= POLY_MASK_START;
          if (punsupported) m |= POLY_MASK_UNSUP;
          st_u64(&DG.help_mask[(uint32_t)tag], m);
        }
      }
      next += (unsigned long long)n_help;
      __syncthreads();
    }
    return;
  }
  const V Q{P, S, P.bkt_head + (size_t)blockIdx.x * 2 * NB * NSUB};
  bool took = false;
  for (;;) {
    int qi;
    if (n_help > 0) {  // (with helpers a leader owns exactly one query: its own index)
      qi = took ? P.nq : (int)blockIdx.x;
      took = true;
    } else {
      if (tid == 0) {
        S.q_index = atomicAdd(P.next_query, 1);
        if (guard_abort(P)) S.q_index = P.nq;  // the host has given up on this launch: take no further query
      }
      __syncthreads();
      qi = S.q_index;
    }
    if (qi >= P.nq) break;
    const int q = P.order[qi];
    const QueryIn &in = P.queries[q];
    const PolyWorld WG = DG.worlds[P.poly_world[q]];
    const unsigned long long t_begin = wall_clock64();
    // the world's obstacles into LDS for the whole search (every collide() walks their trajectories)
    PolyDev D;
    PolyWorld W;
    poly_stage_world<BLOCK>(DG, WG, wlds, tid, D, W);
    if (tid < P.n_u) { pU[tid][0] = D.U[2 * tid]; pU[tid][1] = D.U[2 * tid + 1]; }  // (read after the __syncthreads() that follow)
    for (int i = tid; i < 2 * NB; i += BLOCK) S.cnt[0][i] = 0;
    if (tid == 0) {
      S.n_near = 0; S.n_nodes = 0; S.n_edges = 0; S.n_log = 0;
      S.reserve = (uint32_t)P.n_u;
      S.node_chunks = S.edge_chunks = S.open_chunks = 0;
      S.cur1 = 0; S.cur0 = 0; S.lo1 = 0.0; S.ts_f = INFINITY; S.ts_g = INFINITY; S.ts_id = 0xFFFFFFFFu;
      S.status = -1;
      for (int i = 0; i < 10; i++) S.cyc[i] = 0;
      S.c_expanded = S.c_closed = S.c_prims = S.c_succ = S.c_succ_finite = S.c_reads = 0;
      S.c_push = S.c_reopen = S.c_refill = S.c_evict = 0;
      S.c_hash = 0;
      punsupported = 0;
      plevel[0] = 0ull; plevel[1] = 0ull;
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
      if (S.status < 0) {
        bool ok = ensure_chunks(S.node_tbl, S.node_chunks, 1, NODE_CH_LOG, MAX_NODE_CH, P.chunk_next + 0, P.node_chunks) &&
                  ensure_chunks(S.open_tbl, S.open_chunks, 1, OPEN_CH_LOG, MAX_OPEN_CH, P.chunk_next + 2, P.open_chunks);
        if (!ok) S.status = 4;
      }
    }
    __syncthreads();
    uint32_t goal_id = NIL;
    if (S.status < 0) {
      if (tid == 0) {  // start node (id 0); its key carries the start time
        int32_t key[MAX_KEY];
        state_key_c<CONTROL>(in.start, key);
        key[ns] = (int32_t)round(in.start_t / 0.1);
        char *rec = Q.node(0);
        for (int i = 0; i < NK; i++) V::key(rec)[i] = key[i];
        const double *src = (const double *)&in.start;
        for (int i = 0; i < ns; i++) V::state(rec)[i] = src[i];
        V::state(rec)[ns] = in.start_t;
        double h = P.eps == 0.0 ? 0.0 : get_heur(S.hp, CONTROL, in.start, key, NK);
        V::h(rec) = h;
        V::g(rec) = 0.0;
        V::flags(rec) = FLAG_OPENED;
        V::pred(rec) = NIL;
        const unsigned long long h64 = key_hash64(key, NK);
        const unsigned long long tagq = tbl_tagq(h64, (uint32_t)q, P.tbl_epoch);
        size_t pos = (size_t)(h64 ^ ((unsigned long long)(uint32_t)q * 0x9E3779B97F4A7C15ull)) & (size_t)P.table_mask;
        for (unsigned long long steps = 0;; steps++) {
          const unsigned long long seen = ld_u64(&P.table[pos]);  // (a slot of another epoch is empty: claimed against the value seen)
          if (tbl_empty(seen, P.tbl_epoch) && atomicCAS(&P.table[pos], seen, tagq | 0ull) == seen) break;
          if (steps > P.table_mask) { S.status = 5; break; }  // (the table is full: never with the host's sizing)
          pos = (pos + 1) & (size_t)P.table_mask;
        }
        S.n_nodes = 1;
        S.f_base = 0.0 + P.eps * h;
        S.lo1 = S.f_base;
        S.n_log = 1;
        S.c_push = 1;
      }
      __syncthreads();
      if (tid == 0) open_push(Q, 0u, S.f_base, 0.0, 0u);
      __syncthreads();
      for (;;) {
        while (S.n_near + S.reserve > (uint32_t)NC) {
          evict_half(Q, tid);
          __syncthreads();
        }
        MPLX_TIC(tp);
        const bool popped = pop_min<BLOCK, CONTROL, Smem<BLOCK>, NK>(Q, tid);
        MPLX_TOC(S, 0, tp);
        if (!popped) {
          if (tid == 0) S.status = 1;
          __syncthreads();
          break;
        }
        const uint32_t cur = S.cur_id;
        if (tid == 0) {
          S.c_expanded++;
          S.c_closed++;
          S.c_hash = S.c_hash * 0x100000001B3ull + (unsigned long long)(cur + 1u);
          if (P.rec_ids && S.c_expanded <= P.cap_rec) P.rec_ids[(size_t)q * P.cap_rec + (S.c_expanded - 1)] = (int32_t)cur;
          S.flag = 0;
          pstart_hit = 0;
        }
        // (helpers) the look-ahead record of this state, if a helper has got to it: loaded now, looked at after the primitives
        unsigned long long look = 0ull;
        if (n_help > 0 && tid == 0) look = ld_u64(&DG.help_mask[Q.node_rec(cur)]);
        // ---- env_poly_map::get_succ(curr): S.cur[0] = pos3 vel3 ... , S.cur[0][12] = curr.t
        MPLX_TIC(tx);
        const double T = P.dt, cur_t = S.cur[0][12], t_rel = cur_t - W.start_t;
        LaneSucc L;
        L.valid = false; L.blocked = false; L.reads = 0;
        double lane_cost = 0.0;
        if (tid < P.n_u) {
          const double pos[2] = {S.cur[0][0], S.cur[0][1]}, vel[2] = {S.cur[0][3], S.cur[0][4]}, acc[2] = {S.cur[0][6], S.cur[0][7]}, u[2] = {pU[tid][0], pU[tid][1]};
          double c[2][6];
          poly_prim_build(CONTROL, pos, vel, u, c, acc);
          for (int i = 0; i < 2; i++)
            for (int j = 0; j < 6; j++) pcs[tid][i][j] = c[i][j];
          L.tn.p[0] = pp_p_auto(c[0], T); L.tn.p[1] = pp_p_auto(c[1], T); L.tn.p[2] = 0.0;
          L.tn.v[0] = pp_v_auto(c[0], T); L.tn.v[1] = pp_v_auto(c[1], T); L.tn.v[2] = 0.0;
          for (int k = 0; k < 3; k++) { L.tn.a[k] = 0.0; L.tn.j[k] = 0.0; }
          if constexpr (CONTROL == CTRL_JRK) { L.tn.a[0] = pp_a_auto(c[0], T); L.tn.a[1] = pp_a_auto(c[1], T); }
          pvalid[tid] = (poly_inside(W.bbox, 4, L.tn.p[0], L.tn.p[1]) && poly_validate(CONTROL, c, T, P.v_max, P.a_max, P.j_max)) ? 1 : 0;
          phit[tid] = 0;
          lane_cost = poly_intrinsic_cost(CONTROL, c, T, P.w, P.dt);
          state_key_c<CONTROL>(L.tn, L.key);
          L.key[ns] = (int32_t)round((cur_t + P.dt) / 0.1);
        }
        // first probe of the state table for every valid successor, issued now: its round trip overlaps the collision tests
        unsigned long long h64 = 0, v0 = TBL_EMPTY;
        if (tid < P.n_u && pvalid[tid]) {
          h64 = key_hash64(L.key, NK);
          v0 = ld_u64(&P.table[(size_t)(h64 ^ ((unsigned long long)(uint32_t)q * 0x9E3779B97F4A7C15ull)) & (size_t)P.table_mask]);
        }
        if (n_help > 0 && tid == 0) pmask = look;
        __syncthreads();
        MPLX_TOC(S, 3, tx);
        // isFree(start.pos, t) and isFree(pr, t) of all primitives against all obstacles
        auto hook = [&]() {  // the table slot has arrived: start fetching the record it names (the commit reads it)
                                  const uint32_t vid = (uint32_t)v0;
                                  if (vid < CLAIM_BASE && (v0 >> 32) == (tbl_tagq(h64, (uint32_t)q, P.tbl_epoch) >> 32))
                                    __builtin_prefetch(Q.node(vid), 0, 3);
                                };
        const bool looked = n_help > 0 && (pmask & POLY_MASK_READY) != 0ull;  // (uniform)
        if (looked) {  // a helper has run the collision tests of this state: the same pure function of the state
          const unsigned long long m = pmask;
          if (tid < P.n_u) phit[tid] = (int32_t)((m >> tid) & 1ull);
          if (tid == 0) {
            if (m & POLY_MASK_START) pstart_hit = 1;
            if (m & POLY_MASK_UNSUP) punsupported = 1;
            S.cyc[8]++;
          }
          hook();
          __syncthreads();
        } else {
          poly_collide_all<BLOCK, decltype(hook), GEN>(D, W, pcs, pvalid, P.n_u, T, t_rel, pprep, phit_idx, puns_idx, &php_max, phit, &punsupported, &pstart_hit, tid, (long long)q + 1,
                                hook, S.cyc, plevel);
        }
        if (tid < P.n_u) {
          L.valid = pvalid[tid] != 0;
          L.blocked = L.valid && (pstart_hit || phit[tid]);
        }
        const bool act = L.valid && !L.blocked;
        // (helpers) a state this expansion creates goes into the leader's ring at once: {id | record, pos2, vel2, acc2, t}
        const double t_next = cur_t + P.dt;
        auto publish = [&](uint32_t id, uint32_t recx, const LaneSucc &Ls) {
          if (n_help <= 0) return;
          double *e = DG.help_ring + (((size_t)blockIdx.x << DG.help_ring_log) + ((size_t)id & (((size_t)1 << DG.help_ring_log) - 1))) * 8;
          st_f64x2_agent(e + 0, __longlong_as_double((long long)(((unsigned long long)id << 32) | (unsigned long long)recx)), Ls.tn.p[0]);
          st_f64x2_agent(e + 2, Ls.tn.p[1], Ls.tn.v[0]);
          st_f64x2_agent(e + 4, Ls.tn.v[1], Ls.tn.a[0]);
          st_f64x2_agent(e + 6, Ls.tn.a[1], t_next);
        };
        {
          uint32_t tot;
          block_excl_scan<BLOCK>((L.valid ? 1u : 0u) | (act ? 1u << 10 : 0u), S, tid, tot);
          if (tid == 0) {
            S.c_prims += (unsigned long long)P.n_u;
            S.c_succ += tot & 0x3FFu;
            S.c_succ_finite += tot >> 10;
            if (punsupported) S.status = 5;
          }
        }
        S.dupset[tid] = 0;
        S.dupset[tid + BLOCK] = 0;
        __syncthreads();
        MPLX_TOC(S, 1, tx);
        MPLX_TIC(tc);
        if (S.status >= 0) break;
        if (act) {
          const unsigned long long hv = h64 | 1ull;
          uint32_t sl = (uint32_t)(h64 >> 7) & (2 * BLOCK - 1);
          for (;;) {
            unsigned long long old = atomicCAS(&S.dupset[sl], 0ull, hv);
            if (old == 0ull) break;
            if (old == hv) { S.flag = 1; break; }
            sl = (sl + 1) & (2 * BLOCK - 1);
          }
        }
        __syncthreads();
        if (!S.flag) {
          commit_parallel<BLOCK, CONTROL, Smem<BLOCK>, NK, false, decltype(publish)>(Q, tid, q, act, L, h64, lane_cost, (uint32_t)tid, true, v0, publish);
        } else {
          for (int i = 0; i < P.n_u && S.status < 0; i++)
            commit_parallel<BLOCK, CONTROL, Smem<BLOCK>, NK, false, decltype(publish)>(Q, tid, q, act && tid == i, L, h64, lane_cost, (uint32_t)tid, false, 0ull, publish);
        }
        if (n_help > 0 && tid < 64) {  // the ring entries of the states just created have landed (wave 0 wrote them) before the count names them
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (tid == 0) st_u64(&DG.help_pub[blockIdx.x], (unsigned long long)S.n_nodes);
        }
        __syncthreads();
        MPLX_TOC(S, 2, tc);
        if (S.status >= 0) break;
        if (tid == 0) {
          State s;
          for (int i = 0; i < 12; i++) ((double *)&s)[i] = S.cur[0][i];
          if (S.cur[0][12] >= P.t_max || is_goal_state(s, S.hp.goal, S.hp.goal_control, P.tol_pos, P.tol_vel, P.tol_acc))  // (the LDS copy of the goal)
            S.status = 0;
          else if (P.max_expand > 0 && S.c_expanded >= (unsigned long long)P.max_expand)
            S.status = 3;
          else if ((S.c_expanded & 63ull) == 0ull) {  // launch guard: heartbeat + abort word, every 64th expansion
            guard_mark(P, GUARD_BATCH, (uint32_t)q, S.c_expanded, (unsigned long long)S.n_nodes);
            if (guard_abort(P)) S.status = PLAN_ABORTED;
          }
        }
        __syncthreads();
        if (S.status >= 0) break;
      }
      goal_id = S.cur_id;
      clear_buckets(Q, tid);
    }
    if (n_help > 0 && tid == 0) st_u64(&DG.help_pub[blockIdx.x], POLY_PUB_DONE | (unsigned long long)S.n_nodes);  // the helpers of this leader leave
    __syncthreads();
    if (tid == 0) {  // recoverTraj + results
      QueryOut &o = P.out[q];
      int32_t *tn = P.traj_nodes + (size_t)q * (MAX_TRAJ + 1);
      int32_t *ta = P.traj_actions + (size_t)q * MAX_TRAJ;
      double *ts = P.traj_states + (size_t)q * (MAX_TRAJ + 1) * 13;
      int status = S.status;
      double cost = INFINITY;
      int len = 0;
      auto edge_cost = [&](uint32_t parent, uint32_t action) {  // calculate_intrinsic_cost of Primitive(parent, U[action], dt)
        const double *st = V::state(Q.node(parent));
        const double pos[2] = {st[0], st[1]}, vel[2] = {st[3], st[4]}, acc[2] = {CONTROL == CTRL_JRK ? st[6] : 0.0, CONTROL == CTRL_JRK ? st[7] : 0.0}, u[2] = {D.U[2 * action], D.U[2 * action + 1]};
        double c[2][6];
        poly_prim_build(CONTROL, pos, vel, u, c, acc);
        return poly_intrinsic_cost(CONTROL, c, P.dt, P.w, P.dt);
      };
      if (status == 0 && goal_id == NIL) {
        cost = S.tmp_d0;
      } else if (status == 0) {
        uint32_t node = goal_id;
        tn[0] = (int32_t)node;
        bool ok = true, too_long = false;
        while (V::pred(Q.node(node)) != NIL) {
          uint32_t best = NIL;
          double min_rhs = INFINITY, min_g = INFINITY;
          uint32_t hops = 0;
          for (uint32_t e = V::pred(Q.node(node)); e != NIL && hops <= S.n_edges; e = Q.edge(e)->next, hops++) {
            const EdgeRec er = *Q.edge(e);
            double gp = V::g(Q.node(er.parent));
            double rhs = gp + edge_cost(er.parent, er.action);
            if (rhs < min_rhs || (rhs == min_rhs && gp >= min_g)) { min_rhs = rhs; min_g = gp; best = e; }
          }
          if (best == NIL) { ok = false; break; }
          if (len >= MAX_TRAJ) { too_long = true; break; }
          ta[len] = (int32_t)Q.edge(best)->action;
          node = Q.edge(best)->parent;
          len++;
          tn[len] = (int32_t)node;
          if (node == 0u) break;
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
      o.status = status;
      o.traj_len = len;
      o.cost = cost;
      o.n_expanded = S.c_expanded; o.n_closed = S.c_closed; o.n_nodes = S.n_nodes; o.n_edges = S.n_edges;
      o.n_primitives = S.c_prims; o.n_succ = S.c_succ; o.n_succ_finite = S.c_succ_finite; o.voxel_reads = 0;
      o.n_push = S.c_push; o.n_reopen = S.c_reopen; o.n_refill = S.c_refill; o.n_evict = S.c_evict;
      o.expand_hash = S.c_hash;
      o.n_recorded = (uint32_t)(S.c_expanded < P.cap_rec ? S.c_expanded : P.cap_rec);
      o.slot = blockIdx.x;
      o.spec[0] = o.spec[1] = o.spec[2] = o.spec[3] = 0;
      o.t_begin = t_begin;
      o.t_end = wall_clock64();
      for (int i = 0; i < 10; i++) o.cyc[i] = S.cyc[i];
    }
    for (uint32_t i = tid; i < (uint32_t)MAX_NODE_CH; i += BLOCK)
      P.node_tables[(size_t)q * MAX_NODE_CH + i] = i < S.node_chunks ? S.node_tbl[i] : NIL;
    for (uint32_t i = tid; i < (uint32_t)MAX_EDGE_CH; i += BLOCK)
      P.edge_tables[(size_t)q * MAX_EDGE_CH + i] = i < S.edge_chunks ? S.edge_tbl[i] : NIL;
    __syncthreads();
  }
}


// One successor of env_poly_map::get_succ (mirrors mplx_poly_succ)
struct PolySuccOut {
  double state[9];  // pos2 vel2 acc2 jrk2 t
  double cost;      // intrinsic cost, or +inf when PolyMapUtil::isFree(pr, t) fails
  int32_t action, valid;
};

// env_poly_map::get_succ for K nodes: one workgroup per node.  Lane i < n_u builds primitive i (end state, bounding
// box, validate_primitive, intrinsic cost); then isFree(pr, t) of all of them against all obstacles spread over the lanes
// below the pair level (poly_collide_all, mplx_poly_dev.h) and the start-point test isFree(start.pos, t) over the
// obstacles; results are OR-ed in LDS.
template <int BLOCK, bool GEN = false>
__global__ __launch_bounds__(BLOCK) void poly_get_succ_kernel(PolyDev D, int K, const int32_t *world_of, const double *states, PolySuccOut *out, int32_t *flags) {
  __shared__ double cs[POLY_MAX_U][2][6];
  __shared__ int32_t valid[POLY_MAX_U], hit[POLY_MAX_U];
  __shared__ int32_t start_hit, unsupported, hp_max;
  __shared__ PolyPrep prep[POLY_MAX_OBS];
  __shared__ uint32_t hit_idx[POLY_MAX_U * POLY_MAX_OBS], uns_idx[POLY_MAX_U * POLY_MAX_OBS];
  const int tid = threadIdx.x;
  for (int k = blockIdx.x; k < K; k += gridDim.x) {
    const double *st = states + 9 * (size_t)k;
    const PolyWorld W = D.worlds[world_of[k]];
    const double T = D.dt, t_rel = st[8] - W.start_t;
    if (tid == 0) { start_hit = 0; unsupported = 0; }
    if (tid < D.n_u) {
      const double pos[2] = {st[0], st[1]}, vel[2] = {st[2], st[3]}, acc[2] = {st[4], st[5]}, jrk[2] = {st[6], st[7]}, u[2] = {D.U[2 * tid], D.U[2 * tid + 1]};
      double c[2][6];
      poly_prim_build(D.control, pos, vel, u, c, acc, jrk);
      for (int i = 0; i < 2; i++)
        for (int j = 0; j < 6; j++) cs[tid][i][j] = c[i][j];
      const double ex = pp_p_auto(c[0], T), ey = pp_p_auto(c[1], T);
      valid[tid] = (poly_inside(W.bbox, 4, ex, ey) && poly_validate(D.control, c, T, D.v_max, D.a_max, D.j_max)) ? 1 : 0;
      hit[tid] = 0;
    }
    __syncthreads();
    // isFree(start.pos, t) (start = pr.evaluate(0) = the node position for every primitive) and isFree(pr, t)
    poly_collide_all<BLOCK, PolyNoHook, GEN>(D, W, cs, valid, D.n_u, T, t_rel, prep, hit_idx, uns_idx, &hp_max, hit, &unsupported, &start_hit, tid, 0, PolyNoHook());
    if (tid < D.n_u) {
      PolySuccOut &o = out[(size_t)k * D.n_u + tid];
      double c[2][6];
      for (int a = 0; a < 2; a++)
        for (int b = 0; b < 6; b++) c[a][b] = cs[tid][a][b];
      o.state[0] = pp_p_auto(c[0], T); o.state[1] = pp_p_auto(c[1], T);
      o.state[2] = pp_v_auto(c[0], T); o.state[3] = pp_v_auto(c[1], T);
      o.state[4] = pp_a_auto(c[0], T); o.state[5] = pp_a_auto(c[1], T);
      o.state[6] = pp_j_auto(c[0], T); o.state[7] = pp_j_auto(c[1], T);
      o.state[8] = st[8] + D.dt;
      o.action = tid;
      o.valid = valid[tid];
      o.cost = (start_hit || hit[tid]) ? INFINITY : poly_intrinsic_cost(D.control, c, T, D.w, D.dt);
    }
    if (tid == 0 && unsupported) atomicOr(flags, 1);
    __syncthreads();
  }
}


}  // namespace mplx
