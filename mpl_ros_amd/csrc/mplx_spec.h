// mplx_spec.h -- astar_spec_kernel: the device-resident A* with speculative K-way expansion.
//
// get_succ(curr) is a pure function of (curr, U, dt, limits, map) -- visible in the in-tree sibling
// environments (env_poly_map.h:45-69, env_cloud.h:50-70 read nothing else) -- so expanding several
// OPEN nodes ahead of time is always exact work.  Per iteration the workgroup
//   1. takes the K smallest entries of OPEN (K successive workgroup-wide argmins on the near set),
//   2. expands them concurrently, one expansion unit of UL lanes per node (phases 1-2 of
//      mplx_kernels.h), looks all successors up in the state space and computes the heuristic of
//      the would-be new states -- every global-memory round trip of the batch overlaps,
//   3. commits the units ONE AT A TIME in pop order.  Unit k is committed only if its entry is still
//      smaller (total order f, g, id) than everything pushed by units 0..k-1 of this batch; otherwise
//      the batch is cut and the remaining candidates go back to OPEN untouched.  A small LDS "batch
//      table" carries g / flags / newest-predecessor of every state touched by the batch from unit to
//      unit, so a unit sees exactly the state space the sequential loop would have shown it.
// The pop sequence, node ids, edge order and all results are therefore identical to the sequential
// loop (and to astar_kernel); only wall time changes.  Measured on the CPU oracle, 99.9 % of popped
// nodes were last touched >= 15 expansions earlier, so cuts are rare.
#pragma once
#include "mplx_kernels.h"

namespace mplx {

template <int UL, int K, int CONTROL>
struct SmemSpec : Smem<UL * K, 3 * nq_c(CONTROL), K> {
  static constexpr int BLOCK = UL * K, BT = 2 * UL * K, NK = key_len_c(CONTROL);
  // candidates in pop order
  double cand_f[K], cand_g[K];
  uint32_t cand_id[K], cand_idx[K];
  int32_t cand_live[K];
  int32_t n_cand;
  uint32_t u_succ[K], u_fin[K], u_reads[K];  // per-unit successor / finite-successor / voxel-read totals
  int32_t unit_seq[K];                       // two lanes of the unit share a key -> lane-by-lane commit
  uint32_t cur_slot[K];                      // batch-table slot of node k itself (NIL if not a successor)
  // batch table: one entry per distinct successor key of the batch
  unsigned long long bt_hash[BT];
  uint32_t bt_leader[BT];  // smallest thread index sharing the entry (= first in commit order)
  uint32_t bt_id[BT], bt_flags[BT], bt_pred[BT];
  unsigned long long bt_tslot[BT];
  double bt_g[BT], bt_h[BT];
  int32_t lane_key[UL * K][NK];
  // smallest entry pushed by the units committed so far in this batch
  double mp_f, mp_g;
  uint32_t mp_id;
  int32_t stop;
};

// commit the successors of unit `ku_commit`; `active`: this lane commits now.  Workgroup-uniform.
template <int UL, int K, int CONTROL, class SM, class V>
__device__ __forceinline__ void spec_commit_lanes(const V &Q, SM &S, int tid, int q, int kc, bool active, int my_slot,
                                                  const LaneSucc &L, double hspec) {
  constexpr int BLOCK = UL * K;
  constexpr int nk = key_len_c(CONTROL), ns = key_len_c(CONTROL);
  const SearchParams &P = Q.P;
  const int lu = tid % UL;
  const bool isnew = active && S.bt_id[my_slot] == NIL;
  uint32_t total;
  uint32_t sc = block_excl_scan<BLOCK>((isnew ? 1u : 0u) | (active ? 1u << 12 : 0u), S, tid, total);
  const uint32_t n_new = total & 0xFFFu, n_fin = total >> 12;
  const uint32_t base_nodes = S.n_nodes, base_edges = S.n_edges;
  if (tid == 0) {
    bool ok = ensure_chunks(S.node_tbl, S.node_chunks, base_nodes + n_new, NODE_CH_LOG, MAX_NODE_CH, P.chunk_next + 0, P.node_chunks) &&
              ensure_chunks(S.edge_tbl, S.edge_chunks, base_edges + n_fin, EDGE_CH_LOG, MAX_EDGE_CH, P.chunk_next + 1, P.edge_chunks) &&
              ensure_chunks(S.open_tbl, S.open_chunks, S.n_log + n_fin, OPEN_CH_LOG, MAX_OPEN_CH, P.chunk_next + 2, P.open_chunks);
    if (!ok) S.status = 4;  // MPLX_PLAN_POOL_FULL
  }
  __syncthreads();
  if (S.status >= 0) return;
  bool improved = false;
  double tg = 0.0, hval = 0.0;
  uint32_t id = NIL;
  if (active) {
    char *rec;
    double old_g;
    uint32_t fl, old_pred;
    if (isnew) {  // first arrival at this key: create the state (this lane is the entry's leader)
      id = base_nodes + (sc & 0xFFFu);
      rec = Q.node(id);
      int32_t *kk = V::key(rec);
#pragma unroll
      for (int i = 0; i < nk; i++) kk[i] = L.key[i];
      double *st = V::state(rec);
#pragma unroll
      for (int i = 0; i < ns; i++) st[i] = i < 3 ? L.tn.p[i % 3] : i < 6 ? L.tn.v[i % 3] : i < 9 ? L.tn.a[i % 3] : L.tn.j[i % 3];
      st[ns] = S.cur[kc][12] + P.dt;
      V::h(rec) = hspec;
      S.bt_id[my_slot] = id;
      S.bt_h[my_slot] = hspec;
      const unsigned long long h64 = S.bt_hash[my_slot];
      st_u64(&P.table[S.bt_tslot[my_slot]], (((h64 >> 48) << 48) | ((unsigned long long)(uint32_t)q << 32)) | id);
      old_g = INFINITY;
      fl = 0;
      old_pred = NIL;
      hval = hspec;
    } else {
      id = S.bt_id[my_slot];
      rec = Q.node(id);
      old_g = S.bt_g[my_slot];
      fl = S.bt_flags[my_slot];
      old_pred = S.bt_pred[my_slot];
      hval = S.bt_h[my_slot];
    }
    const uint32_t eidx = base_edges + (sc >> 12);
    EdgeRec *e = Q.edge(eidx);
    e->parent = S.cand_id[kc];
    e->next = old_pred;
    e->action = (uint32_t)lu;
    V::pred(rec) = eidx;
    S.bt_pred[my_slot] = eidx;
    tg = S.cand_g[kc] + P.ucost[lu];
    improved = tg < old_g;
    if (improved) {
      if (fl & FLAG_CLOSED) {  // re-open
        fl &= ~FLAG_CLOSED;
        atomicAdd(&S.c_reopen, 1ull);
        atomicAdd(&S.c_closed, (unsigned long long)-1ll);
      }
      fl |= FLAG_OPENED;
    }
    if (improved || isnew) {
      const double ng = improved ? tg : old_g;
      V::g(rec) = ng;
      V::flags(rec) = fl;
      S.bt_g[my_slot] = ng;
      S.bt_flags[my_slot] = fl;
    }
  }
  uint32_t total_p;
  uint32_t sp = block_excl_scan<BLOCK>(improved ? 1u : 0u, S, tid, total_p);
  const uint32_t base_log = S.n_log;
  double pf = INFINITY, pg = INFINITY;
  uint32_t pi = 0xFFFFFFFFu;
  if (improved) {
    pf = tg + P.eps * hval;
    if (pf != pf) pf = INFINITY;
    pg = tg;
    pi = id;
    open_push(Q, base_log + sp, pf, pg, pi);
  }
  // smallest pushed entry of this commit -> batch minimum
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    double of = __shfl_xor(pf, d, 64), og = __shfl_xor(pg, d, 64);
    uint32_t oi = __shfl_xor(pi, d, 64);
    if (entry_less(of, og, oi, pf, pg, pi)) { pf = of; pg = og; pi = oi; }
  }
  if ((tid & 63) == 0) {
    S.red_f[tid >> 6] = pf;
    S.red_g[tid >> 6] = pg;
    S.red_id[tid >> 6] = pi;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 0; w < BLOCK / 64; w++)
      if (entry_less(S.red_f[w], S.red_g[w], S.red_id[w], S.mp_f, S.mp_g, S.mp_id)) { S.mp_f = S.red_f[w]; S.mp_g = S.red_g[w]; S.mp_id = S.red_id[w]; }
    S.n_nodes = base_nodes + n_new;
    S.n_edges = base_edges + n_fin;
    S.n_log = base_log + total_p;
    S.c_push += total_p;
  }
  __syncthreads();
}

template <int UL, int K, int CONTROL>
__global__ __launch_bounds__(UL *K) void astar_spec_kernel(SearchParams P) {
  constexpr int BLOCK = UL * K;
  using SM = SmemSpec<UL, K, CONTROL>;
  constexpr int BT = SM::BT;
  __shared__ SM S;
  using V = QView<BLOCK, CONTROL, SM>;
  const int tid = threadIdx.x, ku = tid / UL, lu = tid % UL;
  const V Q{P, S, P.bkt_head + (size_t)blockIdx.x * 2 * NB * NSUB};
  constexpr int nk = key_len_c(CONTROL), ns = key_len_c(CONTROL);
  for (;;) {
    if (tid == 0) S.q_index = atomicAdd(P.next_query, 1);
    __syncthreads();
    const int qi = S.q_index;
    if (qi >= P.nq) break;
    const int q = P.order[qi];
    const QueryIn &in = P.queries[q];
    const unsigned long long t_begin = wall_clock64();
    // ---- reset the workgroup's OPEN structure
    for (int i = tid; i < 2 * NB * NSUB; i += BLOCK) Q.bkt_head[i] = NIL;
    for (int i = tid; i < 2 * NB; i += BLOCK) S.cnt[0][i] = 0;
    if (tid == 0) {
      S.n_near = 0; S.n_nodes = 0; S.n_edges = 0; S.n_log = 0;
      S.node_chunks = S.edge_chunks = S.open_chunks = 0;
      S.cur1 = 0; S.cur0 = 0; S.lo1 = 0.0; S.ts_f = INFINITY; S.ts_g = INFINITY; S.ts_id = 0xFFFFFFFFu;
      S.status = -1;
      for (int i = 0; i < 8; i++) S.cyc[i] = 0;
      S.c_expanded = S.c_closed = S.c_prims = S.c_succ = S.c_succ_finite = S.c_reads = 0;
      S.c_push = S.c_reopen = S.c_refill = S.c_evict = 0;
      S.c_hash = 0;
      S.cur_id = NIL;
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
      if (S.status < 0) {
        bool ok = ensure_chunks(S.node_tbl, S.node_chunks, 1, NODE_CH_LOG, MAX_NODE_CH, P.chunk_next + 0, P.node_chunks) &&
                  ensure_chunks(S.open_tbl, S.open_chunks, 1, OPEN_CH_LOG, MAX_OPEN_CH, P.chunk_next + 2, P.open_chunks);
        if (!ok) S.status = 4;
      }
    }
    __syncthreads();
    bool searched = false;
    if (S.status < 0) {
      searched = true;
      // ---- start node (id 0)
      if (tid == 0) {
        int32_t key[MAX_KEY];
        state_key_c<CONTROL>(in.start, key);
        char *rec = Q.node(0);
        for (int i = 0; i < nk; i++) V::key(rec)[i] = key[i];
        const double *src = (const double *)&in.start;
        for (int i = 0; i < ns; i++) V::state(rec)[i] = src[i];
        V::state(rec)[ns] = in.start_t;
        double h = P.eps == 0.0 ? 0.0 : get_heur(S.hp, CONTROL, in.start, key, nk);
        V::h(rec) = h;
        V::g(rec) = 0.0;
        V::flags(rec) = FLAG_OPENED;
        V::pred(rec) = NIL;
        const unsigned long long h64 = key_hash64(key, nk);
        const unsigned long long tagq = ((h64 >> 48) << 48) | ((unsigned long long)(uint32_t)q << 32);
        size_t pos = (size_t)(h64 ^ ((unsigned long long)(uint32_t)q * 0x9E3779B97F4A7C15ull)) & (size_t)P.table_mask;
        for (;;) {
          unsigned long long old = atomicCAS(&P.table[pos], TBL_EMPTY, tagq | 0ull);
          if (old == TBL_EMPTY) break;
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
      // ---- main loop: one batch of up to K expansions per iteration
      for (;;) {
        while (S.n_near > (uint32_t)(NC - BLOCK - K)) {
          MPLX_TIC(te);
          evict_half(Q, tid);
          __syncthreads();
          MPLX_TOC(S, 3, te);
        }
        MPLX_TIC(tp);
        if (S.n_near == 0) {
          __syncthreads();
          if (!refill(Q, tid)) {
            if (tid == 0) S.status = 1;  // OPEN empty
            __syncthreads();
            break;
          }
          __syncthreads();
        }
        // ---- 1. the K smallest OPEN entries, in order
        if (tid == 0) {
          S.cyc[7]++;  // batches
          S.n_cand = 0;
          S.stop = 0;
          S.mp_f = INFINITY; S.mp_g = INFINITY; S.mp_id = 0xFFFFFFFFu;
        }
        if (tid < K) {
          S.cand_live[tid] = 0;
          S.unit_seq[tid] = 0;
          S.cur_slot[tid] = NIL;
          S.u_succ[tid] = S.u_fin[tid] = S.u_reads[tid] = 0;
        }
        __syncthreads();
        for (int k = 0; k < K; k++) {
          const uint32_t n = S.n_near;
          if (n == 0) break;  // uniform
          double bf = INFINITY, bg = INFINITY;
          uint32_t bi = 0xFFFFFFFFu, bp = NIL;
          for (uint32_t i = tid; i < n; i += BLOCK) {
            double f = S.near_f[i], g = S.near_g[i];
            uint32_t id = S.near_id[i];
            if (bp == NIL || entry_less(f, g, id, bf, bg, bi)) { bf = f; bg = g; bi = id; bp = i; }
          }
#pragma unroll
          for (int d = 32; d >= 1; d >>= 1) {
            double of = __shfl_xor(bf, d, 64), og = __shfl_xor(bg, d, 64);
            uint32_t oi = __shfl_xor(bi, d, 64), op = __shfl_xor(bp, d, 64);
            if (op != NIL && (bp == NIL || entry_less(of, og, oi, bf, bg, bi))) { bf = of; bg = og; bi = oi; bp = op; }
          }
          if ((tid & 63) == 0) {
            S.red_f[tid >> 6] = bf; S.red_g[tid >> 6] = bg; S.red_id[tid >> 6] = bi; S.red_pos[tid >> 6] = bp;
          }
          __syncthreads();
          if (tid == 0) {
            for (int w = 1; w < BLOCK / 64; w++) {
              uint32_t op = S.red_pos[w];
              if (op != NIL && (bp == NIL || entry_less(S.red_f[w], S.red_g[w], S.red_id[w], bf, bg, bi))) {
                bf = S.red_f[w]; bg = S.red_g[w]; bi = S.red_id[w]; bp = op;
              }
            }
            S.cand_f[k] = bf; S.cand_g[k] = bg; S.cand_id[k] = bi; S.cand_idx[k] = S.near_idx[bp];
            const uint32_t last = n - 1;
            S.near_f[bp] = S.near_f[last]; S.near_g[bp] = S.near_g[last];
            S.near_id[bp] = S.near_id[last]; S.near_idx[bp] = S.near_idx[last];
            S.n_near = last;
            S.n_cand = k + 1;
          }
          __syncthreads();
        }
        const int n_cand = S.n_cand;
        // ---- 2a. fetch the candidates' records; drop stale entries (improved or closed since pushed)
        bool live_unit = false;
        if (ku < n_cand) {
          char *rec = Q.node(S.cand_id[ku]);
          const double rg = V::g(rec);
          const uint32_t fl = V::flags(rec);
          live_unit = __double_as_longlong(rg) == __double_as_longlong(S.cand_g[ku]) && !(fl & FLAG_CLOSED);
          if (live_unit) {
            if (lu <= ns) S.cur[ku][lu < ns ? lu : 12] = V::state(rec)[lu];
            if (lu >= ns && lu < 12) S.cur[ku][lu] = 0.0;
            if (lu < nk) S.cur_key[ku][lu] = V::key(rec)[lu];
            if (lu == 0) S.cand_live[ku] = 1;
          }
        }
        __syncthreads();
        MPLX_TOC(S, 0, tp);
        // ---- 2b. expand all live units concurrently
        MPLX_TIC(tx);
        LaneSucc L;
        expand_unit<UL, BLOCK, CONTROL>(P, S, tid, live_unit, L);
        const bool act = L.valid && !L.blocked;
        {
          uint32_t tot, treads;
          unit_excl_scan<UL, BLOCK>((L.valid ? 1u : 0u) | (act ? 1u << 10 : 0u), S, tid, tot);
          unit_excl_scan<UL, BLOCK>(L.reads, S, tid, treads);
          if (lu == 0) {
            S.u_succ[ku] = tot & 0x3FFu;
            S.u_fin[ku] = tot >> 10;
            S.u_reads[ku] = treads;
          }
        }
        MPLX_TOC(S, 1, tx);
        // ---- 2c. batch table: one entry per distinct successor key; its leader looks the key up in
        //          the state space (or claims a slot) and computes the heuristic of a new state
        MPLX_TIC(tc);
        unsigned long long h64 = 0;
        int my_slot = 0;
        for (int i = tid; i < BT; i += BLOCK) {
          S.bt_hash[i] = 0ull;
          S.bt_leader[i] = NIL;
        }
        if (act) {
          h64 = key_hash64(L.key, nk);
#pragma unroll
          for (int i = 0; i < nk; i++) S.lane_key[tid][i] = L.key[i];
        }
        __syncthreads();
        if (act) {
          const unsigned long long hv = h64 | 1ull;
          int sl = (int)(h64 >> 7) & (BT - 1);
          for (;;) {
            unsigned long long old = atomicCAS(&S.bt_hash[sl], 0ull, hv);
            if (old == 0ull || old == hv) break;
            sl = (sl + 1) & (BT - 1);
          }
          my_slot = sl;
          atomicMin(&S.bt_leader[sl], (uint32_t)tid);
        }
        __syncthreads();
        double hspec = 0.0;
        if (act) {
          const uint32_t leader = S.bt_leader[my_slot];
          if (leader != (uint32_t)tid) {
            bool eq = true;
#pragma unroll
            for (int i = 0; i < nk; i++) eq = eq && (S.lane_key[leader][i] == L.key[i]);
            if (!eq) S.status = 5;                                    // 64-bit key-hash collision inside a batch
            if ((int)(leader / UL) == ku) S.unit_seq[ku] = 1;       // two lanes of one unit, one key
          } else {
            const unsigned long long tagq = ((h64 >> 48) << 48) | ((unsigned long long)(uint32_t)q << 32);
            const size_t mask = (size_t)P.table_mask;
            size_t pos = (size_t)(h64 ^ ((unsigned long long)(uint32_t)q * 0x9E3779B97F4A7C15ull)) & mask;
            unsigned long long v0 = ld_u64(&P.table[pos]);
            if (P.eps != 0.0) hspec = get_heur(S.hp, CONTROL, L.tn, L.key, nk);  // overlaps the probe
            const unsigned long long claim = tagq | (unsigned long long)(CLAIM_BASE + (uint32_t)tid);
            bool first = true;
            for (;;) {
              unsigned long long v = first ? v0 : ld_u64(&P.table[pos]);
              first = false;
              if (v == TBL_EMPTY) {
                unsigned long long old = atomicCAS(&P.table[pos], TBL_EMPTY, claim);
                if (old == TBL_EMPTY) {
                  S.bt_id[my_slot] = NIL;  // new state; created when its first sharer commits
                  S.bt_tslot[my_slot] = (unsigned long long)pos;
                  S.bt_g[my_slot] = INFINITY;
                  S.bt_flags[my_slot] = 0;
                  S.bt_pred[my_slot] = NIL;
                  break;
                }
                v = old;
              }
              const uint32_t vid = (uint32_t)v;
              if (vid < CLAIM_BASE && (v & 0xFFFFFFFF00000000ull) == tagq) {
                char *r = Q.node(vid);
                const double rg = V::g(r), rh = V::h(r);
                const uint32_t rfl = V::flags(r), rpred = V::pred(r);
                const int32_t *kk = V::key(r);
                bool eq = true;
#pragma unroll
                for (int i = 0; i < nk; i++) eq = eq && (kk[i] == L.key[i]);
                if (eq) {
                  S.bt_id[my_slot] = vid;
                  S.bt_g[my_slot] = rg;
                  S.bt_h[my_slot] = rh;
                  S.bt_flags[my_slot] = rfl;
                  S.bt_pred[my_slot] = rpred;
                  break;
                }
              }
              pos = (pos + 1) & mask;
            }
          }
        }
        // does a candidate itself appear among the successors of the batch?  (its closed flag must
        // reach the units committed after it)
        if (lu == 0 && live_unit) {
          const unsigned long long hk = key_hash64(S.cur_key[ku], nk);
          const unsigned long long hv = hk | 1ull;
          int sl = (int)(hk >> 7) & (BT - 1);
          for (;;) {
            const unsigned long long o = S.bt_hash[sl];
            if (o == 0ull) break;
            if (o == hv) { S.cur_slot[ku] = (uint32_t)sl; break; }
            sl = (sl + 1) & (BT - 1);
          }
        }
        __syncthreads();
        MPLX_TOC(S, 2, tc);
        if (S.status >= 0) break;
        // ---- 3. ordered commit
        MPLX_TIC(to);
        int k_stop = n_cand;
        for (int k = 0; k < n_cand; k++) {
          if (!S.cand_live[k]) continue;  // stale entry: dropped, like a pop that skips it
          if (!entry_less(S.cand_f[k], S.cand_g[k], S.cand_id[k], S.mp_f, S.mp_g, S.mp_id)) {
            k_stop = k;  // something pushed by this batch now precedes candidate k: cut here
            break;
          }
          if (tid == 0) {
            const uint32_t cur = S.cand_id[k];
            char *rec = Q.node(cur);
            V::flags(rec) = V::flags(rec) | FLAG_CLOSED;
            if (S.cur_slot[k] != NIL) S.bt_flags[S.cur_slot[k]] |= FLAG_CLOSED;
            S.c_expanded++;
            S.c_closed++;
            S.c_hash = S.c_hash * 0x100000001B3ull + (unsigned long long)(cur + 1u);
            if (P.rec_ids && S.c_expanded <= P.cap_rec) P.rec_ids[(size_t)q * P.cap_rec + (S.c_expanded - 1)] = (int32_t)cur;
            S.c_prims += (unsigned long long)P.n_u;
            S.c_succ += S.u_succ[k];
            S.c_succ_finite += S.u_fin[k];
            S.c_reads += S.u_reads[k];
            S.cur_id = cur;
          }
          __syncthreads();
          if (!S.unit_seq[k]) {
            spec_commit_lanes<UL, K, CONTROL>(Q, S, tid, q, k, act && ku == k, my_slot, L, hspec);
          } else {
            for (int i = 0; i < P.n_u && S.status < 0; i++)
              spec_commit_lanes<UL, K, CONTROL>(Q, S, tid, q, k, act && ku == k && lu == i, my_slot, L, hspec);
          }
          if (S.status >= 0) { k_stop = k + 1; break; }  // pool full
          if (tid == 0) {
            State s;
            for (int i = 0; i < 12; i++) ((double *)&s)[i] = S.cur[k][i];
            if (S.cur[k][12] >= P.t_max || is_goal_state(s, in.goal, in.goal_control, P.tol_pos, P.tol_vel, P.tol_acc))
              S.status = 0;
            else if (P.max_expand > 0 && S.c_expanded >= (unsigned long long)P.max_expand)
              S.status = 3;
          }
          __syncthreads();
          if (S.status >= 0) { k_stop = k + 1; break; }
        }
        __syncthreads();
        // candidates behind a cut go back to OPEN untouched
        if (tid == 0 && S.status < 0) {
          for (int k = k_stop; k < n_cand; k++) {
            if (!S.cand_live[k]) continue;
            const uint32_t pos = S.n_near++;
            S.near_f[pos] = S.cand_f[k]; S.near_g[pos] = S.cand_g[k]; S.near_id[pos] = S.cand_id[k]; S.near_idx[pos] = S.cand_idx[k];
          }
        }
        __syncthreads();
        MPLX_TOC(S, 6, to);
        if (S.status >= 0) break;
      }
    }
    const uint32_t goal_id = searched ? S.cur_id : NIL;
    __syncthreads();
    // ---- recoverTraj + results (thread 0)
    if (tid == 0) {
      QueryOut &o = P.out[q];
      int32_t *tn = P.traj_nodes + (size_t)q * (MAX_TRAJ + 1);
      int32_t *ta = P.traj_actions + (size_t)q * MAX_TRAJ;
      double *ts = P.traj_states + (size_t)q * (MAX_TRAJ + 1) * 13;
      int status = S.status;
      double cost = INFINITY;
      int len = 0;
      if (status == 0 && goal_id == NIL) {
        cost = S.tmp_d0;
      } else if (status == 0) {
        uint32_t node = goal_id;
        tn[0] = (int32_t)node;
        bool ok = true;
        while (V::pred(Q.node(node)) != NIL) {
          uint32_t best = NIL;
          double min_rhs = INFINITY, min_g = INFINITY;
          for (uint32_t e = V::pred(Q.node(node)); e != NIL; e = Q.edge(e)->next) {
            const EdgeRec er = *Q.edge(e);
            double gp = V::g(Q.node(er.parent));
            double rhs = gp + P.ucost[er.action];
            if (rhs < min_rhs || (rhs == min_rhs && gp >= min_g)) { min_rhs = rhs; min_g = gp; best = e; }
          }
          if (best == NIL || len >= MAX_TRAJ) { ok = false; break; }
          ta[len] = (int32_t)Q.edge(best)->action;
          node = Q.edge(best)->parent;
          len++;
          tn[len] = (int32_t)node;
          if (node == 0u) break;
        }
        if (ok) {
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
      o.n_primitives = S.c_prims; o.n_succ = S.c_succ; o.n_succ_finite = S.c_succ_finite; o.voxel_reads = S.c_reads;
      o.n_push = S.c_push; o.n_reopen = S.c_reopen; o.n_refill = S.c_refill; o.n_evict = S.c_evict;
      o.expand_hash = S.c_hash;
      o.n_recorded = (uint32_t)(S.c_expanded < P.cap_rec ? S.c_expanded : P.cap_rec);
      o.slot = blockIdx.x;
      o.t_begin = t_begin;
      o.t_end = wall_clock64();
      for (int i = 0; i < 8; i++) o.cyc[i] = S.cyc[i];
    }
    for (uint32_t i = tid; i < (uint32_t)MAX_NODE_CH; i += BLOCK)
      P.node_tables[(size_t)q * MAX_NODE_CH + i] = i < S.node_chunks ? S.node_tbl[i] : NIL;
    __syncthreads();
  }
}

}  // namespace mplx
