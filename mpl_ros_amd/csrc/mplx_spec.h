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
// loop (and to astar_kernel); only wall time changes.  Measured on a CPU run of the same search, 99.9 % of popped
// nodes were last touched >= 15 expansions earlier, so cuts are rare.
#pragma once
#include "mplx_kernels.h"

#ifdef MPLX_LOOKUP_TIMERS
#define MPLX_T2(S, k, var) do { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); (S).cyc2[k] += now_ - (var); var = now_; } } while (0)
#else
#define MPLX_T2(S, k, var) do { } while (0)
#endif

// Switches of the leader's batch loop (tools/build_kernel_variant.sh builds A/B variants with -DMPLX_X_...=0); the values
// here are the product configuration.
#ifndef MPLX_X_LDS_CONST
#define MPLX_X_LDS_CONST 1    // control inputs, edge costs in LDS; goal test against the LDS copy of the goal
#endif
#ifndef MPLX_X_EARLY_SETUP
#define MPLX_X_EARLY_SETUP 1  // next batch's set-up (resets, chunk capacity, helper announcement) inside the end-of-batch bookkeeping
#endif
#ifndef MPLX_X_EARLY_ROW
#define MPLX_X_EARLY_ROW 1    // look-ahead cache row (heuristics, voxel-read count) requested as soon as the row is known
#endif
#ifndef MPLX_X_EARLY_TOMB
#define MPLX_X_EARLY_TOMB 0   // (A/B, off) TBL_DEAD_ID of an abandoned claim stored ahead of the parallel commit instead of behind it
#endif
#ifndef MPLX_X_ROW_FENCE
#define MPLX_X_ROW_FENCE 0    // look-ahead rows (units of 32 lanes) published behind an agent-scope RELEASE fence (buffer_wbl2 sc1 + s_waitcnt) instead of validated by a check word
#endif
// run-time twins of the A/B switches (SearchParams::xflags, MPLX_X_FLAGS): 16 row fence, 32 early tombstone, 64 claim wait in the one-node kernels
#define MPLX_ROW_FENCE(P) (MPLX_X_ROW_FENCE || MPLX_XF(P, 16))
#define MPLX_EARLY_TOMB(P) (MPLX_X_EARLY_TOMB || MPLX_XF(P, 32))
#ifndef MPLX_X_EARLY_CLEAR
#define MPLX_X_EARLY_CLEAR 1  // batch table cleared by the idle waves of the end-of-batch bookkeeping
#endif
// Measured in round 6 and NOT kept (profiles/r06_ab_negative_results.json; every variant returned identical results):
//   * settling a pending far-bucket link only when the next batch walks a far list: tail 2032 vs 2028 ms, bulk 136.1 vs 134.6 ms;
//   * helpers that prefer a leader of their own XCD and publish rows / records with plain stores (kept in the shared L2 instead of
//     written through): one long query with four same-XCD helpers 2057-2084 vs 2030 ms, the blocking C4 step 2516 vs 2301 ms (helpers
//     wait for a same-XCD leader while others go unserved) -- the hand-over's trips to memory are not what bounds a batch;
//   * 6 / 8 voxel loads in flight per lane in the sampling loop instead of 4: bulk 139.0 / 148.4 vs 134.6 ms (17 / 33 spilled VGPRs);
//   * examining 2 K OPEN entries per batch (each unit's two lane halves fetch one record, staged through LDS) and expanding the first K
//     LIVE ones, so that stale entries -- 13 % of the candidates -- do not cost a unit its slot: identical results, tail 2273 vs 2027 ms,
//     bulk 139.6 vs 135.2 ms (profiles/r06h_ab_refill_negative.json): one more barrier, 27 spilled VGPRs, and a fresh push inside the
//     first 2 K entries shifts every later one away from the record its half unit had prefetched -- an exposed refetch in most batches.
#ifndef MPLX_X_PROBE2
#define MPLX_X_PROBE2 1       // (round 6) the look-up's first load brings the home slot AND its neighbour (almost always the same 64-byte line): a
                              // key whose home slot is taken by another state no longer costs the slowest lane of the batch a second dependent trip
#endif

namespace mplx {

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not drain
// outstanding global-memory operations (vmcnt), so a returning global atomic issued before it
// keeps overlapping.  Used where units hand data to each other through LDS exclusively.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// YAW: yaw-carrying states -- one more key integer (round(yaw / 0.1)) behind the control kind's own
template <int UL, int K, int CONTROL, int BTN, int NCAP_, bool YAW = false>
struct SmemSpec : Smem<UL * K, K, NCAP_> {
  static constexpr int BLOCK = UL * K, BT = BTN, NK = key_len_c(CONTROL) + (YAW ? 1 : 0);
  static_assert(NK <= MAX_KEY, "yaw-carrying SNP states (13 key integers) stay on the one-node kernel");
  // candidates in pop order
  double cand_f[K], cand_g[K];
  uint32_t cand_id[K], cand_idx[K], cand_pos[K];
  int32_t cand_live[K];
  uint32_t cand_fl[K];  // node flags of candidate k when its record was fetched
  int32_t n_cand;
  uint32_t u_succ[K], u_fin[K], u_reads[K];  // per-unit successor / finite-successor / voxel-read totals
  int32_t unit_seq[K];                       // two lanes of the unit share a key -> lane-by-lane commit
  uint32_t cur_slot[K];                      // batch-table slot of node k itself (NIL if not a successor)
  // batch table: one entry per distinct successor key of the batch
  unsigned long long bt_hash[BT];
  uint32_t bt_leader[BT];  // smallest thread index sharing the entry (= first in commit order)
  uint32_t bt_id[BT], bt_flags[BT], bt_pred[BT], bt_dirty[BT];
  uint32_t bt_share[BT];   // lanes besides the leader that reach the entry: count << 16 | (sum of their thread indices); bit 31: a candidate of the batch
  // (potential-field searches) the per-lane potential sums of the expansion live in bt_share, which is idle until the
  // batch table is rebuilt after the expansion (BT >= BLOCK lanes)
  __device__ __forceinline__ uint32_t *pot_scratch() { static_assert(BT >= UL * K, "one word per lane"); return bt_share; }
  uint32_t bt_tslot[BT];  // slot of the state table claimed for a new state (the host keeps tables at <= 2^32 slots)
  double bt_g[BT], bt_h[BT];
  int32_t lane_key[UL * K][NK];
  int32_t u_goal[K];  // candidate k satisfies the goal test (evaluated ahead of its commit)
  int32_t u_cut[K];   // first later candidate preceded by an entry unit k pushes (K if none)
  uint32_t n_sorted;  // near_[0, n_sorted) is in ascending order (left so by the previous selection)
  // the entries appended since, sorted (selection scratch)
  unsigned long long hpow[17];  // powers of the expansion-hash multiplier (0x100000001B3^e mod 2^64), filled once per workgroup
  double app_f[256], app_g[256];
  uint32_t app_id[256], app_rank[256];
  uint32_t csum[2][UL * K / 64];  // per-wave totals of the parallel commit's workgroup scan, alternating between batches
  int32_t batch_dep;  // units interact through a state one of them MODIFIES -> ordered, unit-by-unit commit
  int32_t any_shared; // some state is reached by two lanes (they only append predecessor edges unless batch_dep)
  int32_t dep_cause;  // (debug statistics) 1 shared successor, 2 candidate is a successor, 4 a sharer modifies the state
  int32_t cut_at;     // first candidate preceded by an entry pushed in this batch (K if none)
  // helper workgroups (mplx_device.h): leader side
  int32_t helped;               // a helper is attached to this workgroup's box (sampled every few batches)
  unsigned long long box_seq;   // wish lists published for the running query
  // helper side
  int32_t help_box, help_idx, help_q, help_go, help_quit;
#ifdef MPLX_HELP_DEBUG
  unsigned long long dbg_t, dbg_gap, dbg_when;
#endif
  unsigned long long help_seq;
  int32_t help_idle;
  uint32_t n_work;
  uint32_t work[WISH];          // pool indices of the node records to expand ahead of time
  unsigned long long wish_l[WISH];
#ifdef MPLX_LOOKUP_TIMERS
  unsigned long long cyc2[24];
  unsigned long long cycw[16][4];
  unsigned long long arr_max, arr_heur, sum_heur, sum_arr;
  unsigned long long dbg_n, dbg_na, dbg_slow, dbg_n256, dbg_pulls;
#endif
};

// Commit the successors of candidate `kc`; `active`: this lane commits now (all active lanes belong
// to unit kc).  Chunk capacity for the whole batch was reserved before the ordered loop, ids come
// from a scan inside the unit, and the only workgroup barrier is the one the caller places between
// two units.  A far-bucket link is left pending in (pend_idx, pend_old): the caller stores
// open(pend_idx)->next = pend_old after the ordered loop, so the atomicExch latency overlaps it.
struct LanePre {   // per-lane values of the ordered commit that do not depend on the other units
  double tg, pf;   // tentative g through this candidate, f of the entry it would push
  int code;        // where that entry goes (near / fine / coarse bucket)
  int cut;         // first later candidate such an entry would precede (K if none)
};

template <int UL, int K, int CONTROL, bool PAR, bool HELP, bool YAW, class SM, class V>
__device__ __forceinline__ void spec_commit_lanes(const V &Q, SM &S, int tid, int q, int kc, bool active, int my_slot, int k_stop,
                                                  const LaneSucc &L, double hspec, const LanePre &pre, uint32_t &pend_idx, uint32_t &pend_old, int par = 0) {
  constexpr int BLOCK = UL * K;
  constexpr int nk = key_len_c(CONTROL), ns = key_len_c(CONTROL);
  const SearchParams &P = Q.P;
  const int lu = tid % UL;
  const bool isnew = active && S.bt_id[my_slot] == NIL;
  if (active && pend_idx != NIL) {  // lane-by-lane mode can reach here twice: settle the earlier link
    Q.open(pend_idx)->next = pend_old;
    pend_idx = NIL;
  }
  double old_g = INFINITY, tg = 0.0;
  uint32_t fl = 0, old_pred = NIL, id = NIL;
  bool improved = false;
  if (active) {
    if (!isnew) {
      id = S.bt_id[my_slot];
      old_g = S.bt_g[my_slot];
      fl = S.bt_flags[my_slot];
      old_pred = S.bt_pred[my_slot];
    }
    tg = pre.tg;
    improved = tg < old_g;
  }
  // The pool counters this commit starts from are read HERE, ahead of the scan's barrier: the lane that publishes the new totals
  // at the end of this function belongs to ONE wave, and nothing but that barrier separates its store from the other waves'
  // reads.  [Until round 5 the three reads stood behind the scan: a wave that fell behind its publisher -- a unit of 128 lanes is
  // two waves, the parallel commit is all of them; a memory pipeline under back-pressure is what makes a wave fall behind --
  // started from the UPDATED counters: its states, predecessor and log records landed past the new totals (holes of unwritten
  // records behind them, the records themselves overwritten by the next batch).  Found in round 5 on the 1024-query jerk batch
  // (<128,4,JRK,help>: 20-60 queries differing from run to run, runs of 7-110 all-zero node records in their pools; the
  // helper-less build of the same source happened to be scheduled the other way round), and the likeliest origin of round 4's
  // "states created twice under a background fill load" as well.]
  const uint32_t base_nodes = S.n_nodes, base_edges = S.n_edges, base_log = S.n_log;
  // PAR: every committing unit at once (units proven independent), ids from a workgroup scan in
  // (unit, lane) order = the order the unit-by-unit loop would assign
  uint32_t total;
  const uint32_t packed = (isnew ? 1u : 0u) | (active ? 1u << 10 : 0u) | (improved ? 1u << 20 : 0u);
  uint32_t sc;
  if constexpr (PAR) {
    // workgroup scan with ONE barrier: the per-wave totals go to the half of csum this batch owns (the other half may
    // still be read by a wave that is late leaving the previous batch's scan -- many barriers ago)
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t x = wave_incl_sum<64>(packed);
    if (lane == 63) S.csum[par][opaque(wave)] = x;
    lds_barrier();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; w++) {
      const uint32_t t = S.csum[par][w];
      if (w < wave) base += t;
      tot += t;
    }
    total = tot;
    sc = base + x - packed;
  } else {
    sc = unit_excl_scan<UL, BLOCK>(packed, S, tid, total);
  }
  uint32_t chain_next = old_pred;
  bool write_pred = true;
  if (PAR && S.any_shared) {  // (uniform) two lanes append an edge to one state: chain them in thread order
    const uint32_t share = active ? S.bt_share[my_slot] : 0u;
    const bool two = ((share >> 16) & 0x7FFFu) == 1u;
    const bool lead = two && S.bt_leader[my_slot] == (uint32_t)tid;
    if (lead && (int)((share & 0xFFFFu) / UL) < k_stop) {  // the other lane commits too and becomes the newest predecessor
      S.bt_pred[my_slot] = base_edges + ((sc >> 10) & 0x3FFu);
      write_pred = false;
    }
    lds_barrier();
    if (two && !lead) chain_next = S.bt_pred[my_slot];
  }
  if (active) {
    if (isnew) {  // first arrival at this key: create the state (this lane is the entry's leader)
      id = base_nodes + (sc & 0x3FFu);
      char *rec = Q.node(id);
      int32_t *kk = V::key(rec);
#pragma unroll
      for (int i = 0; i < nk; i++) kk[i] = L.key[i];
      if constexpr (YAW) kk[nk] = L.yaw_key;
      double *st = V::state(rec);
      if (HELP && !MPLX_XF(P, 512)) {  // helper workgroups on other compute units read the state: agent-scope (write-through) stores,
        // 16 bytes at a time (the state starts on a 64-byte boundary of the record)   [MPLX_X_FLAGS & 512, diagnostics: plain stores]
        auto sv = [&](int i) { return i < 3 ? L.tn.p[i % 3] : i < 6 ? L.tn.v[i % 3] : i < 9 ? L.tn.a[i % 3] : L.tn.j[i % 3]; };
#pragma unroll
        for (int i = 0; i + 1 < ns; i += 2) st_f64x2_agent(&st[i], sv(i), sv(i + 1));
        if constexpr (ns & 1) st_f64_agent(&st[ns - 1], sv(ns - 1));
      } else {
#pragma unroll
        for (int i = 0; i < ns; i++) st[i] = i < 3 ? L.tn.p[i % 3] : i < 6 ? L.tn.v[i % 3] : i < 9 ? L.tn.a[i % 3] : L.tn.j[i % 3];
      }
      if constexpr (YAW) {  // the state's yaw sits between the control kind's own doubles and t (the one-node kernel's layout)
        st[ns] = L.yaw;
        st[ns + 1] = S.cur[kc][12] + P.dt;
      } else {
        st[ns] = S.cur[kc][12] + P.dt;
      }
      V::h(rec) = hspec;
      S.bt_id[my_slot] = id;
      S.bt_h[my_slot] = hspec;
      const unsigned long long h64 = S.bt_hash[my_slot];
      st_u64(&P.table[S.bt_tslot[my_slot]], tbl_tagq(h64, (uint32_t)q, P.tbl_epoch) | id);
    }
    const uint32_t eidx = base_edges + ((sc >> 10) & 0x3FFu);
    EdgeRec *e = Q.edge(eidx);
    e->parent = S.cand_id[kc];
    e->next = chain_next;
    e->action = (uint32_t)lu | (L.pot << EDGE_POT_SHIFT);  // control input | potential sum of the primitive's samples (0 without a potential field)
    if (PAR) {
      if (write_pred) V::pred(Q.node(id)) = eidx;  // nobody else modifies this state in the batch
    } else {
      S.bt_pred[my_slot] = eidx;
      S.bt_dirty[my_slot] = 1;  // g / flags / newest predecessor reach the record once, after the ordered loop
    }
    if (improved) {
      if (fl & FLAG_CLOSED) {  // re-open
        fl &= ~FLAG_CLOSED;
        atomicAdd(&S.c_reopen, 1ull);
        atomicAdd(&S.c_closed, (unsigned long long)-1ll);
      }
      fl |= FLAG_OPENED;
    }
    if (improved || isnew) {
      if (PAR) {
        char *rec2 = Q.node(id);
        V::g(rec2) = improved ? tg : old_g;
        V::flags(rec2) = fl;
      } else {
        S.bt_g[my_slot] = improved ? tg : old_g;
        S.bt_flags[my_slot] = fl;
      }
    }
    if (improved) {
      const double pf = pre.pf;
      const uint32_t idx = base_log + (sc >> 20);
      OpenRec *r = Q.open(idx);
      r->f = pf;
      r->g = tg;
      r->id = id;
      const int code = pre.code;
      if (code < 0) {
        uint32_t pos = atomicAdd(&S.n_near, 1u);
        S.near_f[pos] = pf; S.near_g[pos] = tg; S.near_id[pos] = id; S.near_idx[pos] = idx;
      } else {
        const uint32_t c = atomicAdd(&S.cnt[0][code], 1u);  // (sub-list by the bucket's running count: see far_link)
        pend_old = atomicExch(&Q.bkt_head[(size_t)code * NSUB + (c & (NSUB - 1))], idx);
        pend_idx = idx;
      }
      if (!PAR && pre.cut < K) atomicMin(&S.cut_at, pre.cut);  // this entry precedes a later candidate: cut there
    }
  }
  // the unit's first lane publishes the new totals (in lane-by-lane mode that lane may be inactive:
  // totals come from the scan, which every lane of the unit sees)
  if (PAR ? tid == 0 : tid == kc * UL) {
    S.n_nodes = base_nodes + (total & 0x3FFu);
    S.n_edges = base_edges + ((total >> 10) & 0x3FFu);
    S.n_log = base_log + (total >> 20);
    S.c_push += total >> 20;
  }
}

// ------------------------------------------------------------------ helper workgroups (look-ahead expansion)
// A workgroup that has no query left to lead turns into a helper (the tail of astar_spec_kernel<..., HELP>): there is
// no second launch and nothing waiting for a compute unit, so whatever takes the launch's waves off the machine and
// puts them back (the driver evicts and restores a process's queues around memory-management events) finds room
// for all of them again.  [An earlier version ran the helpers as a second launch on a second stream, more workgroups
// than compute units, waiting to be dispatched as leaders exited: after an eviction the waiting ones took the
// leaders' compute units and the leaders stayed off the machine until those helpers gave up -- seconds.]
// A helper leaves when every query is done or the cache is full; when it has found nobody to help HELP_IDLE_ROUNDS
// times in a row (leaders that want help are there from the start: a helper that finds every leader served is
// surplus); or -- a safety net, the leader never waits for a helper -- when the leader it serves has not completed a
// batch for HELP_STALL_POLLS polls.  Both limits count the helper's own iterations, not wall-clock time: time spent
// off the machine is not progress missed.
constexpr int HELP_IDLE_ROUNDS = 1000;     // x ~54 us
constexpr int HELP_STALL_POLLS = 100000;   // x >= 3.4 us (13.6 us after the first few)
#ifdef MPLX_HELP_DEBUG
__device__ __forceinline__ uint32_t dbg_xcc() {  // XCD this wave runs on (0..7)
  uint32_t x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 15u;
}
#endif
// Check word of a look-ahead cache row (units of 32 lanes: upper half of the row's voxel-read slot).  The record that names a row is
// written after the row "has landed" (s_waitcnt vmcnt(0) on the helper's side) -- but the acknowledgement of a posted agent-scope
// store is not a promise that a reader on another XCD sees it before a LATER store to another line: under a write-heavy
// neighbour (another launch's hipMemset) a leader was seen consuming the record and then the row's previous contents (wrong
// heuristics: another expansion order, or an OPEN list that runs dry).  The row therefore validates itself end to end: XOR of a
// term per heuristic the leader will read, the voxel-read count, and a salt of (state key, query, launch epoch) that no other
// row carries; a leader that reads a row whose check word does not match reads it again until it does.
__device__ __forceinline__ uint32_t cache_row_term(double h, int lu) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(h);
  return (((uint32_t)b ^ (uint32_t)(b >> 32)) + (uint32_t)lu) * 0x9E3779B1u;
}
__device__ __forceinline__ uint32_t cache_row_salt(uint32_t khash, uint32_t q, uint32_t epoch, uint32_t reads) {
  return khash ^ (q * 0x85EBCA77u) ^ (epoch * 0xC2B2AE3Du) ^ (reads * 0x27D4EB2Fu) ^ 0xA5A5A5A5u;
}
__device__ __forceinline__ uint32_t unit32_xor(uint32_t x) {  // XOR over the 32 lanes of a unit (half a wave), in every lane
#pragma unroll
  for (int d = 16; d >= 1; d >>= 1) x ^= (uint32_t)__shfl_xor((int)x, d, 32);
  return x;
}
constexpr uint32_t CACHE_ROW_POLLS = 1u << 18;
// MPLX_X_ROW_PAIRS: check of ONE value of a row (the bits of a heuristic, or the voxel-read count), bound to the state, the query,
// the launch and the slot
__device__ __forceinline__ unsigned long long cache_pair_tag(unsigned long long bits, uint32_t khash, uint32_t q, uint32_t epoch, uint32_t slot) {
  return (bits * 0x9E3779B97F4A7C15ull) ^ (((unsigned long long)khash << 32) | (unsigned long long)q) ^
         ((unsigned long long)epoch * 0xC2B2AE3D27D4EB4Full) ^ ((unsigned long long)(slot + 1u) << 56) ^ 0x5A5A5A5A5A5A5A5Aull;
}

// Serve the leader of box `bi` until its query ends: every time it announces a wish list, expand the listed
// nodes that have no cache entry yet -- get_succ (phases 1-2 of expand_unit) plus the heuristic of every
// finite successor -- and publish {row, voxel reads, valid mask, blocked mask} in cache_c and the heuristics in
// a row of cache_h.  Ordering: the row is written (agent scope) and drained (vmcnt 0) before the two
// self-validating halves of the cache record; the leader reads the record first and the row after it.
template <int UL, int K, int CONTROL, class SM>
__device__ __forceinline__ void helper_serve(const SearchParams &P, SM &S, int tid) {
  constexpr int BLOCK = UL * K;
  constexpr int nk = key_len_c(CONTROL), ns = key_len_c(CONTROL);
  using V = QView<BLOCK, CONTROL, SM>;
  const int ku = tid / UL, lu = tid % UL;
  HelpBox *const B = P.boxes + S.help_box;
  const uint32_t q = (uint32_t)S.help_q;
  const uint32_t epoch = P.epoch;
  unsigned long long last_seq = ((unsigned long long)epoch << 32) | 1ull;
  for (;;) {
    if (tid == 0) {
      unsigned long long seq;
      int go = 1;
      for (int spin = 0;; spin++) {
        seq = ld_u64(&B->seq);
        if (!box_active(seq, epoch) || ld_u32(&B->q) != q || ld_u32(P.cache_next) >= P.cache_rows) { go = 0; break; }
        if (seq != last_seq) break;
        if ((spin & 255) == 255) {  // launch guard: the host has given up on this launch
          guard_mark(P, GUARD_HELPER, q, (unsigned long long)spin, (unsigned long long)S.help_box);
          if (guard_abort(P)) { S.help_quit = 1; go = 0; break; }
        }
        // poll gently: every few microseconds (a leader's batch takes ~25)
        for (int z = 0; z < (spin < 8 ? 1 : 4); z++) __builtin_amdgcn_s_sleep(127);
        if (spin > HELP_STALL_POLLS) {  // the leader makes no progress: leave the launch
          S.help_quit = 1;
          go = 0;
#ifdef MPLX_HELP_DEBUG
          atomicAdd(P.cache_next + 128 + (dbg_xcc() & 7u) * 8u + (((uint32_t)B->xcc_plus1 - 1u) & 7u), 1u);
#endif
          break;
        }
      }
      S.help_go = go;
      S.help_seq = seq;
    }
    __syncthreads();
    if (!S.help_go) break;
    last_seq = S.help_seq;
    // ---- the announced list -> the entries that are mine and still missing
    if (tid < WISH) {
      const unsigned long long v = ld_u64(&B->wish[((uint32_t)last_seq - 2u) & 1u][tid]);
      bool ok = v != ~0ull && (uint32_t)(v >> 48) == (q & 0xFFFFu);  // (16 bits of the query index: enough to tell a stale list from the running query's)
      const uint32_t rec = (uint32_t)(v & 0xFFFFFFFFFFFFull);
      if (ok && P.help_max > 1) {
        const uint32_t m = ld_u32(&B->helpers);
        const int nh = __popc(m);  // several helpers: split by record index (this helper's rank among those attached)
        if (nh > 1) ok = (int)(rec % (uint32_t)nh) == __popc(m & ((1u << S.help_idx) - 1u));
      }
      if (ok) ok = (uint32_t)ld_u64((const unsigned long long *)&P.cache_c[rec]) == 0u;
      const unsigned long long mk = __ballot(ok);
      if (ok) S.work[__popcll(mk & ((1ull << tid) - 1ull))] = rec;
      if (tid == 0) S.n_work = (uint32_t)__popcll(mk);
    }
    __syncthreads();
    const uint32_t n_work = S.n_work;
    for (uint32_t base = 0; base < n_work; base += K) {
      const bool live = base + (uint32_t)ku < n_work;
      const uint32_t rec = live ? S.work[base + ku] : 0u;
      if (live) {
        const double *st = V::state(P.node_pool + (size_t)rec * rec_bytes(CONTROL));
        if (lu < ns) S.cur[ku][lu] = ld_f64_agent(&st[lu]);
        if (lu >= ns && lu < 13) S.cur[ku][lu] = 0.0;
      }
      if (lu == 0) S.hc_row[ku] = 0;
      unit_sync<UL>();
      if (live && lu == 0) {
        State sc;
        for (int i = 0; i < 12; i++) ((double *)&sc)[i] = S.cur[ku][i];
        state_key_c<CONTROL>(sc, S.cur_key[ku]);
      }
      unit_sync<UL>();
      LaneSucc L;
      expand_unit<UL, BLOCK, CONTROL>(P, S, tid, live, L);
      const bool act = L.valid && !L.blocked;
      double h = 0.0;
      if (act && P.eps != 0.0) h = get_heur(S.hp, CONTROL, L.tn, L.key, nk);
      uint32_t treads;
      unit_excl_scan<UL, BLOCK>(L.reads, S, tid, treads);
      const unsigned long long bv = __ballot(L.valid), bb = __ballot(L.blocked);
      uint32_t vmask = 0, bmask = 0;
      if constexpr (UL == 32) {
        vmask = (tid & 32) ? (uint32_t)(bv >> 32) : (uint32_t)bv;
        bmask = (tid & 32) ? (uint32_t)(bb >> 32) : (uint32_t)bb;
      } else if constexpr (UL == 64) {
        vmask = (uint32_t)bv;
        bmask = (uint32_t)bb;
      }
      if (live && lu == 0) {
        const uint32_t row = atomicAdd(P.cache_next, 1u);
        S.hc_row[ku] = row < P.cache_rows ? row + 1u : 0u;
      }
      unit_sync<UL>();
      const uint32_t rp1 = live ? S.hc_row[ku] : 0u;
#if MPLX_X_ROW_PAIRS
      if constexpr (UL == 32) {  // {value, its check}: one 16-byte store per pair
        const uint32_t kh = (uint32_t)key_hash64(S.cur_key[ku], nk);
        double *row = P.cache_h + (size_t)(rp1 - 1u) * cache_row_doubles(UL);
        if (rp1 && lu < P.n_u)
          st_f64x2_agent(&row[cache_h_slot(UL, lu)], h, __longlong_as_double((long long)cache_pair_tag((unsigned long long)__double_as_longlong(h), kh, q, epoch, (uint32_t)lu)));
        if (rp1 && lu == UL - 1)
          st_f64x2_agent(&row[cache_reads_slot(UL)], __longlong_as_double((long long)(unsigned long long)treads),
                         __longlong_as_double((long long)cache_pair_tag((unsigned long long)treads, kh, q, epoch, 63u)));
      } else
#endif
      if (rp1 && lu < P.n_u) st_f64_agent(&P.cache_h[(size_t)(rp1 - 1u) * cache_row_doubles(UL) + cache_h_slot(UL, lu)], h);
      unsigned long long reads_word = (unsigned long long)treads;
      if constexpr (UL == 32 && !MPLX_X_ROW_PAIRS) {  // check word of the row (cache_row_term above)
        const uint32_t cs = unit32_xor((act && P.eps != 0.0) ? cache_row_term(h, lu) : 0u) ^
                            cache_row_salt((uint32_t)key_hash64(S.cur_key[ku], nk), q, epoch, treads);
        reads_word |= (unsigned long long)cs << 32;
      }
      if (rp1 && lu == UL - 1 && !(UL == 32 && MPLX_X_ROW_PAIRS)) st_u64((unsigned long long *)&P.cache_h[(size_t)(rp1 - 1u) * cache_row_doubles(UL) + cache_reads_slot(UL)], reads_word);
      if constexpr (UL > 64) {  // large lattice: every wave of the unit leaves its two words of each mask in the row
        if (rp1 && (tid & 63) == 0) {
          uint32_t *rw = (uint32_t *)(P.cache_h + (size_t)(rp1 - 1u) * cache_row_doubles(UL));
          const int w2 = 2 * (lu >> 6);
          st_u32(rw + w2, (uint32_t)bv); st_u32(rw + w2 + 1, (uint32_t)(bv >> 32));
          st_u32(rw + 4 + w2, (uint32_t)bb); st_u32(rw + 4 + w2 + 1, (uint32_t)(bb >> 32));
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the row has landed before the record names it
      // (large lattices: the row carries the masks as well and has no check word yet -- a full agent-scope release instead)
      if (UL > 64 || MPLX_XF(P, 2) || MPLX_ROW_FENCE(P)) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      unit_sync<UL>();
      if (rp1 && lu == 0) {
        unsigned long long *cr = (unsigned long long *)&P.cache_c[rec];
        st_u64(cr + 1, ((unsigned long long)bmask << 32) | (unsigned long long)(vmask | CACHE_READY));
        st_u64(cr, ((unsigned long long)(uint32_t)key_hash64(S.cur_key[ku], nk) << 32) | (unsigned long long)rp1);
      }
      if (lu == 0) S.hc_row[ku] = 0;
      __syncthreads();
    }
  }
}

// A workgroup with no query to lead: attach to the longest-running leader that lacks a helper, serve it until
// its query ends, repeat until every query of the batch is done.
template <int UL, int K, int CONTROL, class SM>
__device__ __forceinline__ void helper_loop(const SearchParams &P, SM &S, int tid) {
  constexpr int BLOCK = UL * K;
  const int nboxes = P.help_lead;  // boxes of the workgroups that lead queries
  for (;;) {
    if (tid == 0) {
      // leave when every query is done or the cache is full, when there has been nobody to help for a while, or
      // when the leader just served stopped making progress (helper_serve sets help_quit)
      // (with a helper limit -- streamed batches -- a helper that finds nobody to serve leaves after ~3 ms instead of ~50: its
      // compute unit is wanted by the next batch)
      const bool expired = S.help_quit || S.help_idle > (P.help_limit >= 0 ? 64 : HELP_IDLE_ROUNDS);
      if (S.help_quit) atomicAdd(P.cache_next + 2, 1u);  // (diagnostics) helpers that left a leader that had stopped
      else if (expired) atomicAdd(P.cache_next + 3, 1u);  // (diagnostics) helpers that found every leader served
      const unsigned long long dw = ld_u64(P.done_word);
      const bool done = (uint32_t)(dw >> 32) == P.epoch && (uint32_t)dw >= (uint32_t)P.nq;
      S.flag = (expired || done || ld_u32(P.cache_next) >= P.cache_rows || ((S.help_idle & 15) == 15 && guard_abort(P))) ? 1 : 0;
    }
    __syncthreads();
    if (S.flag) return;
    // whom to help: the leader that has been expanding the longest (in steps of 65 536 expansions, ~0.1 s); among
    // those -- at the start of a batch: everybody -- the query predicted longest (earliest in the launch order)
    unsigned long long best = 0;  // ((steps + 1) << 20) | (2^20 - 1 - min(rank, 2^20 - 1)) < 2^53
    int bi = -1;
    for (int b = tid; b < nboxes; b += BLOCK) {
      const HelpBox *B = P.boxes + b;
      if (!box_active(ld_u64(&B->seq), P.epoch)) continue;
      if (__popc(ld_u32(&B->helpers)) >= P.help_max) continue;
      const uint32_t rk = ld_u32(&B->rank);
      const unsigned long long key = (((ld_u64(&B->n_expanded) >> 16) + 1ull) << 20) | (unsigned long long)(0xFFFFFu - (rk < 0xFFFFFu ? rk : 0xFFFFFu));
      if (key > best) { best = key; bi = b; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const unsigned long long ob = __shfl_xor(best, d, 64);
      const int oi = __shfl_xor(bi, d, 64);
      if (ob > best || (ob == best && oi >= 0 && (bi < 0 || oi < bi))) { best = ob; bi = oi; }
    }
    if ((tid & 63) == 0) {
      S.red_f[tid >> 6] = (double)best;  // (exact: the key is below 2^53)
      S.red_id[tid >> 6] = (uint32_t)bi;
    }
    __syncthreads();
    if (tid == 0) {
      double bb = 0.0;
      int pick = -1;
      for (int w = 0; w < BLOCK / 64; w++)
        if ((int)S.red_id[w] >= 0 && S.red_f[w] > bb) { bb = S.red_f[w]; pick = (int)S.red_id[w]; }
      S.help_box = -1;
      if (pick >= 0) {
        HelpBox *B = P.boxes + pick;
        for (int hh = 0; hh < P.help_max; hh++) {
          const uint32_t old = atomicOr(&B->helpers, 1u << hh);
          if (!(old & (1u << hh))) { S.help_box = pick; S.help_idx = hh; break; }
        }
        if (S.help_box >= 0) {
          const uint32_t q = ld_u32(&B->q);
          if (q < (uint32_t)P.nq && box_active(ld_u64(&B->seq), P.epoch)) {
#ifdef MPLX_HELP_DEBUG
            atomicAdd(P.cache_next + 192 + (dbg_xcc() & 7u) * 8u + (((uint32_t)B->xcc_plus1 - 1u) & 7u), 1u);
#endif
            const QueryIn &in = P.queries[q];
            S.help_q = (int)q;
            S.hp.w = P.w; S.hp.v_max = P.v_max; S.hp.heur_ignore_dynamics = P.heur_ignore_dynamics;
            S.hp.goal_control = in.goal_control;
            S.hp.goal = in.goal;
            S.hp.goal_nkey = state_key(in.goal_control, in.goal, S.hp.goal_key);
          } else {
            atomicAnd(&B->helpers, ~(1u << S.help_idx));
            S.help_box = -1;
          }
        }
      }
    }
    __syncthreads();
    if (S.help_box < 0) {  // nobody to help right now: stay off the memory system for a while
      if (tid == 0) S.help_idle++;
      for (int i = 0; i < 16; i++) __builtin_amdgcn_s_sleep(127);
      __syncthreads();
      continue;
    }
    helper_serve<UL, K, CONTROL>(P, S, tid);
    if (tid == 0) {
      atomicAnd(&(P.boxes + S.help_box)->helpers, ~(1u << S.help_idx));
      S.help_idle = 0;  // the idle count restarts after useful work
    }
    __syncthreads();
  }
}

// POT: the auxiliary map (potential field / search region, MapDev::aux) is read next to the occupancy and the edge cost is
// ucost + pot_weight * (sum of the potential over the primitive's samples): per lane, carried in the predecessor record
// like in the one-node kernel.  Without helper workgroups (their cache rows carry no potential sums).
// YAW (round 4): yaw-carrying states (use_yaw lattices, map_planner_node.cpp:119-139,165) on the speculative kernel: one more
// key integer and one more state double through candidate fetch, batch table, table look-up and creation; the rules are the
// one-node kernel's (astar_kernel<..., YAW>: successor yaw and validate_yaw in expand_unit, heuristic of the yaw-less search
// unless the yaw keys differ from the goal's, optional yaw tolerance in the goal test).  Without helper workgroups.
// FILTER (round 5): a candidate is expanded only if SearchParams::filter_* says so -- the Dijkstra of getSubStateSpace walks only the
// states that had been expanded in the space it leaves; a candidate the filter rejects is dropped like a stale entry.
template <int UL, int K, int CONTROL, int BTN, int NCAP_, bool HELP = false, bool POT = false, bool YAW = false, bool FILTER = false>
__global__ __launch_bounds__(UL *K) void astar_spec_kernel(SearchParams P) {
  static_assert(!(FILTER && (HELP || POT || YAW)), "the filtered search is the plain kernel");
  static_assert(!(HELP && POT), "the look-ahead cache rows carry no potential sums");
  static_assert(!(YAW && (HELP || POT)), "yaw-carrying searches run without helpers and without an auxiliary map");
  constexpr int BLOCK = UL * K;
  using SM = SmemSpec<UL, K, CONTROL, BTN, NCAP_, YAW>;
  constexpr int BT = SM::BT;
  __shared__ SM S;
  using V = QView<BLOCK, CONTROL, SM>;
  const int tid = threadIdx.x, ku = tid / UL, lu = tid % UL;
  // HELP: workgroups with no query left to lead turn into helpers (helper_loop above); a leader publishes the front
  // of its OPEN list and picks up the look-ahead cache entries they leave.  Compiled out of the plain variant (the
  // kernel sits at the register limit).
  const V Q{P, S, P.bkt_head + (size_t)blockIdx.x * 2 * NB * NSUB};
  constexpr int nk = key_len_c(CONTROL), ns = key_len_c(CONTROL);
  constexpr int NKY = nk + (YAW ? 1 : 0), EX = YAW ? 1 : 0;  // key integers / extra state doubles (the yaw) of a record
  // key of a lane's successor incl. the yaw key, its 64-bit hash
  auto lane_hash = [&](const LaneSucc &l) {
    if constexpr (YAW) {
      int32_t kk[MAX_KEY + 1];
#pragma unroll
      for (int i = 0; i < nk; i++) kk[i] = l.key[i];
      kk[nk] = l.yaw_key;
      return key_hash64(kk, NKY);
    } else {
      return key_hash64(l.key, nk);
    }
  };
  // is_goal with the optional yaw tolerance of a yaw-carrying search (astar_kernel's goal_reached)
  auto goal_reached = [&](const State &s, double yaw) {
    bool g = is_goal_state(s, S.hp.goal, S.hp.goal_control & 15, P.tol_pos, P.tol_vel, P.tol_acc);
    if (YAW && g && P.tol_yaw >= 0) g = fabs(yaw - S.hp.goal_yaw) <= P.tol_yaw;
    return g;
  };
  // A node's state doubles: the HELP builds write them with agent-scope (write-through) stores, for the helper workgroups -- and
  // such a store does not refresh a copy of the line in this compute unit's own L1 (round 3 found that for the table slots).  A record
  // whose size is not a multiple of the L1 line (160-byte jerk-state records, 128-byte lines) shares its first line with the tail of
  // the record before it: reading THAT record's last doubles pulls the line in, and when the neighbour is created afterwards -- at a
  // batch boundary -- a plain load of its position returns what the pool held before (another query's state of the previous launch,
  // or garbage: a sample count of billions, i.e. a launch that "hangs").  Round 5 found 34-37 of the 1024 queries of the C4-JRK batch
  // differing from run to run that way; the 128-byte records of the ACC builds are line-aligned and were never exposed.  The leader
  // therefore reads state doubles past the L1 whenever they were stored past it.  (MPLX_X_FLAGS & 128: the old plain loads, to show
  // the difference.)
  auto ld_state = [&](const double *p) -> double {
    if constexpr (HELP) {
      if (!MPLX_XF(P, 128)) return ld_f64_agent(p);
    }
    return *p;
  };
  fill_uq<BLOCK, CONTROL>(P, S, tid);
  if ((P.xflags & 8) && blockIdx.x == 0 && tid == 0) {  // (tests: a launch that does not end by itself -- the host's deadline must)
    guard_mark(P, GUARD_TEST_HANG, 0u, 0ull, 0ull);
    while (!guard_abort(P)) __builtin_amdgcn_s_sleep(127);
  }
  if (tid == 0) {
    unsigned long long pw = 1ull;
    for (int e = 0; e < 17; e++) { S.hpow[e] = pw; pw *= 0x100000001B3ull; }
  }
  for (;;) {
    if (tid == 0) {
      if (HELP && (int)blockIdx.x >= P.help_lead) S.q_index = P.nq;  // a workgroup launched to help only (batch smaller than the machine)
      else {
        S.q_index = atomicAdd(P.next_query, 1);
        if (guard_abort(P)) S.q_index = P.nq;  // launch guard: the host has given up on this launch, take no further query
      }
    }
    __syncthreads();
    const int qi = S.q_index;
    if (qi >= P.nq) break;
    const int q = P.order[qi];
    const QueryIn &in = P.queries[q];
    const unsigned long long t_begin = wall_clock64();
    // ---- reset the workgroup's OPEN structure
    for (int i = tid; i < 2 * NB; i += BLOCK) S.cnt[0][i] = 0;
    if (tid == 0) {
      S.n_near = 0; S.n_nodes = 0; S.n_edges = 0; S.n_log = 0;
      S.reserve = (uint32_t)(K * P.n_u + K);
      S.n_sorted = 0;
      S.node_chunks = S.edge_chunks = S.open_chunks = 0;
      S.cur1 = 0; S.cur0 = 0; S.lo1 = 0.0; S.ts_f = INFINITY; S.ts_g = INFINITY; S.ts_id = 0xFFFFFFFFu;
      S.status = -1;
      for (int i = 0; i < 10; i++) S.cyc[i] = 0;
#ifdef MPLX_LOOKUP_TIMERS
      for (int i = 0; i < 24; i++) S.cyc2[i] = 0;
      for (int i = 0; i < 64; i++) (&S.cycw[0][0])[i] = 0;
      S.arr_max = 0; S.arr_heur = 0; S.sum_heur = 0; S.sum_arr = 0;
      S.dbg_n = S.dbg_na = S.dbg_slow = S.dbg_n256 = S.dbg_pulls = 0;
#endif
      S.c_expanded = S.c_closed = S.c_prims = S.c_succ = S.c_succ_finite = S.c_reads = 0;
      S.c_push = S.c_reopen = S.c_refill = S.c_evict = 0;
      S.c_hash = 0;
      S.c_cand = S.c_live = S.c_cut = 0;
      S.cur_id = NIL;
      S.helped = 0;
      S.box_seq = 0;
      if constexpr (HELP) {  // announce the query (helpers filter wish entries by q, so the order of the stores is free)
        HelpBox *box = P.boxes + blockIdx.x;
        st_u32(&box->q, (uint32_t)q);
        st_u32(&box->rank, (uint32_t)qi);
#ifdef MPLX_HELP_DEBUG
        box->xcc_plus1 = dbg_xcc() + 1u;
        S.dbg_t = wall_clock64();
        S.dbg_gap = 0;
        S.dbg_when = 0;
#endif
        st_u64(&box->n_expanded, 0ull);
        st_u64(&box->seq, ((unsigned long long)P.epoch << 32) | 1ull);
      }
      S.hp.w = P.w; S.hp.v_max = P.v_max; S.hp.heur_ignore_dynamics = P.heur_ignore_dynamics;
      S.hp.goal_control = in.goal_control;
      S.hp.goal = in.goal;
      S.hp.goal_nkey = state_key(in.goal_control, in.goal, S.hp.goal_key);
      S.hp.goal_yaw = in.goal_yaw;
      S.hp.goal_yaw_key = (int32_t)round(in.goal_yaw / KEY_RES_YAW);
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
      else if (in.start_t >= P.t_max || goal_reached(in.start, in.start_yaw)) {
        S.status = 0;
        cost0 = 0.0;
      }
      S.tmp_d0 = cost0;
      if (S.status < 0) {
        bool ok = Q.ensure_nodes(1) &&
                  Q.ensure_open(1);
        if (!ok) S.status = 4;
      }
    }
    __syncthreads();
    bool searched = false;
    if (S.status < 0) {
      searched = true;
      // ---- start node (id 0)
      if (tid == 0) {
        int32_t key[MAX_KEY + 1];
        state_key_c<CONTROL>(in.start, key);
        const int32_t ykey = (int32_t)round(in.start_yaw / KEY_RES_YAW);
        if (YAW) key[nk] = ykey;
        char *rec = Q.node(0);
        for (int i = 0; i < NKY; i++) V::key(rec)[i] = key[i];
        const double *src = (const double *)&in.start;
        for (int i = 0; i < ns; i++) {
          if constexpr (HELP) st_f64_agent(&V::state(rec)[i], src[i]); else V::state(rec)[i] = src[i];
        }
        if (YAW) V::state(rec)[ns] = in.start_yaw;
        V::state(rec)[ns + EX] = in.start_t;
        double h = 0.0;
        if (P.eps != 0.0) h = (YAW && ykey != S.hp.goal_yaw_key) ? cal_heur(S.hp, CONTROL, in.start) : get_heur(S.hp, CONTROL, in.start, key, nk);
        V::h(rec) = h;
        V::g(rec) = 0.0;
        V::flags(rec) = FLAG_OPENED;
        V::pred(rec) = NIL;
        const unsigned long long h64 = key_hash64(key, NKY);
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
      // ---- main loop: one batch of up to K expansions per iteration
      // far-bucket link whose atomicExch (an HBM round trip) is still in flight: open(pend_idx)->next = pend_old is
      // stored at the top of the NEXT iteration -- nothing walks a far list before that -- so the round trip
      // overlaps the end-of-batch bookkeeping instead of being waited for
      uint32_t pend_idx = NIL, pend_old = NIL;
      uint32_t batch_no = 0;  // (parity: which half of csum the batch's commit scan uses)
#if MPLX_X_EARLY_SETUP
      // Set-up of a batch, run inside the end-of-batch bookkeeping of the batch before it (and once before the first):
      // the workgroup is waiting for thread 0's counters there anyway, so the resets (wave 0, after its own reads of what
      // they reset), the chunk capacity for everything the batch can create (one pool per wave 1-3) and the sampling of
      // the helper flag (an agent-scope load, every eighth batch) cost nothing on the chain, and the barrier that used
      // to follow them at the head of the batch is gone.  Pool exhaustion is noticed one barrier earlier than before and
      // ends the query at the same point (nothing of the batch that lacked room was ever committed).
      auto batch_setup = [&](unsigned long long n_expanded_now) {
        // (the per-unit words -- cand_live, unit_seq, cur_slot, u_goal, u_cut, hc_row -- are reset by each unit's first
        // lane in 2a, before anybody reads them; u_succ / u_fin / u_reads are written by every unit after the expansion)
        if ((tid & 63) == 0 && tid >= 64 && tid < 256) {  // (thread 0 is busy with the counters: the other waves' first lanes)
          if (tid == 64) {
            if (S.status < 0) S.cyc[7]++;  // batches
            S.n_cand = 0;
            S.cut_at = K;
            S.batch_dep = 0;
            S.any_shared = 0;
            S.dep_cause = 0;
            if constexpr (HELP) {
              // announce the wish list written during the batch that just ended (complete: a __syncthreads() lies
              // between); the expansion count is the helpers' hint for choosing whom to serve (one batch old here)
              if (S.helped && S.box_seq) {
                HelpBox *box = P.boxes + blockIdx.x;
                st_u64(&box->n_expanded, n_expanded_now);
                st_u64(&box->seq, ((unsigned long long)P.epoch << 32) | (S.box_seq + 1ull));
              }
            }
          }
          // chunk capacity for everything the batch can create: one pool per wave
          const uint32_t room = (uint32_t)(K * P.n_u + K);
          bool ok;
          if (tid == 64)
            ok = Q.ensure_nodes(S.n_nodes + room);
          else if (tid == 128)
            ok = Q.ensure_edges(S.n_edges + room);
          else
            ok = Q.ensure_open(S.n_log + room);
          if (!ok && S.status < 0) S.status = 4;  // MPLX_PLAN_POOL_FULL
          if constexpr (HELP) {  // (the helper flag: an agent-scope load, every eighth batch)
            if (tid == 192 && (S.cyc[7] & 7ull) <= 1ull) S.helped = ld_u32(&(P.boxes + blockIdx.x)->helpers) != 0u;
          }
          // launch guard: heartbeat (posted store) and the host's abort word (a load over the fabric), every 256th batch (~ 4 ms), on
          // a wave that only waits for thread 0's counters here
          if (tid == 192 && (batch_no & 255u) == 0u) {  // (batch_no: this thread's own count of the query's batches)
            guard_mark(P, GUARD_BATCH, (uint32_t)q, (unsigned long long)batch_no, S.c_expanded);
            if (guard_abort(P) && S.status < 0) S.status = PLAN_ABORTED;
          }
        }
#if MPLX_X_EARLY_CLEAR
        if constexpr (!POT) {  // the batch table, by the upper half of the workgroup (idle here)
          if (tid >= BLOCK / 2) {
            for (int i = opaque(tid - BLOCK / 2); i < BT; i += BLOCK / 2) {
              S.bt_hash[i] = 0ull;
              S.bt_leader[i] = NIL;
              S.bt_dirty[i] = 0;
              S.bt_share[i] = 0;
            }
          }
        }
#endif
      };
      batch_setup(0ull);
      __syncthreads();
#endif
      for (;;) {
        if (pend_idx != NIL) {
          Q.open(pend_idx)->next = pend_old;
          pend_idx = NIL;
        }
        while (S.n_near + S.reserve > (uint32_t)SM::NCAP) {
          MPLX_TIC(te);
          evict_half(Q, tid);
          if (tid == 0) S.n_sorted = 0;
          __syncthreads();
          MPLX_TOC(S, 3, te);
        }
        MPLX_TIC(tp);
        [[maybe_unused]] unsigned long long t3 = __builtin_readcyclecounter();
        batch_no++;
        if (S.n_near == 0) {
          __syncthreads();
          if (!refill(Q, tid)) {
            if (tid == 0) S.status = 1;  // OPEN empty
            __syncthreads();
            break;
          }
          if (tid == 0) S.n_sorted = 0;  // a refill may split the near set
          __syncthreads();
        }
        // top up: a batch wants K entries; pulling the next far bucket early keeps the structure exact
        const unsigned long long evict0 = S.c_evict;
        for (int guard = 0; guard < 8 && S.n_near < (uint32_t)K; guard++) {
          __syncthreads();
          const uint32_t before = S.n_near;
          if (!refill(Q, tid)) break;
          if (tid == 0 && S.c_evict != evict0) S.n_sorted = 0;  // only an eviction inside the pull reorders; appends keep the prefix
          (void)before;
          __syncthreads();
        }
        MPLX_T2(S, 16, t3);
        // ---- 1. the K smallest OPEN entries, in order
#if MPLX_X_EARLY_SETUP
        // (the batch's set-up -- resets, chunk capacity, helper announcement -- was done by the end-of-batch bookkeeping
        // of the previous batch, or by the prologue: batch_setup below; no barrier here)
#ifdef MPLX_HELP_DEBUG
        if (HELP && tid == 0) {  // longest time between two batch starts of this slot, when (since the query began), of which query, helped or not
          HelpBox *box = P.boxes + blockIdx.x;
          const unsigned long long now = wall_clock64(), gap = now - S.dbg_t;
          S.dbg_t = now;
          if (gap > S.dbg_gap) { S.dbg_gap = gap; S.dbg_when = ((now - t_begin) << 1) | (S.helped ? 1ull : 0ull); }
          if (gap > 1000000ull) {  // > 10 ms: leave a record right away
            if (gap > box->pad1[0]) { box->pad1[0] = gap; box->pad1[1] = S.dbg_when; box->pad1[2] = (unsigned long long)q; box->pad1[3] = S.cyc[7]; }
          }
        }
#endif
        MPLX_T2(S, 17, t3);
#else
        if (tid == 0) {
          S.cyc[7]++;  // batches
          S.n_cand = 0;
          S.cut_at = K;
          S.batch_dep = 0;
          S.any_shared = 0;
          S.dep_cause = 0;
          if constexpr (HELP) {
            HelpBox *box = P.boxes + blockIdx.x;
#ifdef MPLX_HELP_DEBUG
            {  // longest time between two batch starts of this slot, when (since the query began), of which query, helped or not
              const unsigned long long now = wall_clock64(), gap = now - S.dbg_t;
              S.dbg_t = now;
              if (gap > S.dbg_gap) { S.dbg_gap = gap; S.dbg_when = ((now - t_begin) << 1) | (S.helped ? 1ull : 0ull); }
              if (gap > 1000000ull) {  // > 10 ms: leave a record right away
                if (gap > box->pad1[0]) { box->pad1[0] = gap; box->pad1[1] = S.dbg_when; box->pad1[2] = (unsigned long long)q; box->pad1[3] = S.cyc[7]; }
              }
            }
#endif
            // announce the wish list written during the previous batch (complete: a __syncthreads() lies between)
            if (S.helped && S.box_seq) {
              st_u64(&box->n_expanded, S.c_expanded);
              st_u64(&box->seq, ((unsigned long long)P.epoch << 32) | (S.box_seq + 1ull));
            }
            if ((S.cyc[7] & 7ull) == 1ull) S.helped = ld_u32(&box->helpers) != 0u;
          }
        }
        if ((tid & 63) == 0 && tid < 192) {  // chunk capacity for everything this batch can create: one pool per wave
          const uint32_t room = (uint32_t)(K * P.n_u + K);
          bool ok;
          if (tid == 0)
            ok = Q.ensure_nodes(S.n_nodes + room);
          else if (tid == 64)
            ok = Q.ensure_edges(S.n_edges + room);
          else
            ok = Q.ensure_open(S.n_log + room);
          if (!ok) S.status = 4;  // MPLX_PLAN_POOL_FULL
        }
        if (tid < K) {
          S.cand_live[tid] = 0;
          S.unit_seq[tid] = 0;
          S.cur_slot[tid] = NIL;
          S.u_succ[tid] = S.u_fin[tid] = S.u_reads[tid] = 0;
          S.u_goal[tid] = 0;
          S.u_cut[tid] = K;
          if constexpr (HELP) S.hc_row[tid] = 0;
        }
        __syncthreads();
        MPLX_T2(S, 17, t3);
#endif
        // Look-ahead cache records of the LIKELY candidates, issued before the ranking so that their (agent-scope,
        // always-missing) loads overlap it: the head of the sorted prefix is almost always what the ranking selects
        // (fresh pushes rarely enter the top K); a candidate that turns out different is looked up in 2a as before.
        // The same goes for the candidates' node records: unit ku fetches the record of the ku-th entry of the
        // sorted prefix now and keeps it in registers across the ranking (nothing is committed in between).
        [[maybe_unused]] unsigned long long pf_a = 0, pf_b = 0;
        uint32_t pf_id = NIL;
        double pf_g = 0.0, pf_s = 0.0;
        uint32_t pf_fl = 0;
        int32_t pf_k = 0;
        if ((uint32_t)ku < S.n_sorted && S.n_sorted <= S.n_near) {
          pf_id = S.near_id[ku];
          char *prec = Q.node(pf_id);
          pf_g = V::g(prec);
          pf_fl = V::flags(prec);
          if (lu <= ns + EX) pf_s = ld_state(&V::state(prec)[lu]);
          if (lu < NKY) pf_k = V::key(prec)[lu];
          if constexpr (HELP) {
            if (S.helped && lu == UL - 1) {
              const unsigned long long *cr = (const unsigned long long *)&P.cache_c[Q.node_rec(pf_id)];
              pf_a = ld_u64(cr);
              pf_b = ld_u64(cr + 1);
            }
          }
        }
        {
          // rank every near entry among all of them (strict total order -> unique ranks): ranks
          // 0..K-1 are the candidates in pop order, the others move to position rank-K (which also
          // leaves the near set sorted).  No serial section.
          const uint32_t n = S.n_near;
          const uint32_t kc = n < (uint32_t)K ? n : (uint32_t)K;
          constexpr int PERT = (SM::NCAP + BLOCK - 1) / BLOCK;
          double ef[PERT], eg[PERT];
          uint32_t ei[PERT], ex[PERT], rk[PERT];
#pragma unroll
          for (int r = 0; r < PERT; r++) {
            const uint32_t i = tid + r * BLOCK;
            const bool v = i < n;
            ef[r] = v ? S.near_f[i] : INFINITY;
            eg[r] = v ? S.near_g[i] : INFINITY;
            ei[r] = v ? S.near_id[i] : 0xFFFFFFFFu;
            ex[r] = v ? S.near_idx[i] : NIL;
            rk[r] = 0;
          }
          // near_[0, ns) is sorted (previous selection), near_[ns, n) are the entries appended since
          const uint32_t ns_ = S.n_sorted <= n ? S.n_sorted : 0u;
          const uint32_t na = n - ns_;
#ifdef MPLX_LOOKUP_TIMERS
          if (tid == 0) { S.dbg_n += n; S.dbg_na += na; S.dbg_slow += na > 256u; S.dbg_n256 += n > 256u; }
#endif
          if (na <= 256u) {  // (uniform) the usual case: the entries pushed by the previous batch (a deep query: 100-200)
            // sort the appended entries among themselves (all pairs, G threads per entry), then every
            // entry finds the number of appended entries before it with a binary search
            const uint32_t G = na <= 64u ? BLOCK / 64 : na <= 128u ? BLOCK / 128 : BLOCK / 256;
            const uint32_t a = (uint32_t)tid / G, part = (uint32_t)tid % G;
            uint32_t cnt = 0;
            double af = 0.0, ag = 0.0;
            uint32_t ai = 0;
            if (a < na) {
              af = S.near_f[ns_ + a]; ag = S.near_g[ns_ + a]; ai = S.near_id[ns_ + a];
              for (uint32_t j = part; j < na; j += G)
                if (entry_less(S.near_f[ns_ + j], S.near_g[ns_ + j], S.near_id[ns_ + j], af, ag, ai)) cnt++;
            }
            // the G consecutive lanes of an entry add up their counts; lane `part == 0` (the only one that uses it) ends
            // up with the total (row_shl:n = 0x100 + n: lane i receives lane i + n of its row of 16, or nothing)
            if (G > 1u) cnt += dpp_u32<0x101, 0xf>(0u, cnt);
            if (G > 2u) cnt += dpp_u32<0x102, 0xf>(0u, cnt);
            if (G > 4u) cnt += dpp_u32<0x104, 0xf>(0u, cnt);
            static_assert(BLOCK / 64 <= 8, "at most 8 lanes per appended entry");
            if (a < na && part == 0) {
              S.app_f[cnt] = af; S.app_g[cnt] = ag; S.app_id[cnt] = ai;
              S.app_rank[a] = cnt;
            }
            lds_barrier();  // (LDS only: the record loads issued above stay in flight across the whole ranking)
            // one binary search per entry, in the OTHER sorted sequence (a prefix entry among the appended ones, an
            // appended entry in the prefix), all in the same loop: the lanes of a wave hold both kinds, and two
            // separate loops would run one after the other; the entries of a thread are independent chains
            static_assert(PERT <= 2, "the merged search below handles two entries per thread");
            struct Search { const double *f, *g; const uint32_t *id; uint32_t lo, hi; };
            auto begin = [&](int r) {
              const uint32_t i = tid + r * BLOCK;
              const bool app = i >= ns_;
              return Search{app ? S.near_f : S.app_f, app ? S.near_g : S.app_g, app ? S.near_id : S.app_id, 0u, i < n ? (app ? ns_ : na) : 0u};
            };
            auto advance = [&](Search &b, double mf, double mg, uint32_t mi) {
              if (b.lo < b.hi) {
                const uint32_t mid = (b.lo + b.hi) >> 1;
                if (entry_less(b.f[mid], b.g[mid], b.id[mid], mf, mg, mi)) b.lo = mid + 1; else b.hi = mid;
              }
            };
            Search b0 = begin(0), b1 = begin(PERT - 1);
            if (PERT == 1) b1.hi = 0;
            while (__any(b0.lo < b0.hi || b1.lo < b1.hi)) {
              advance(b0, ef[0], eg[0], ei[0]);
              if constexpr (PERT > 1) advance(b1, ef[PERT - 1], eg[PERT - 1], ei[PERT - 1]);
            }
            {
              const uint32_t i = tid;
              if (i < ns_) rk[0] = i + b0.lo; else if (i < n) rk[0] = b0.lo + S.app_rank[i - ns_];
            }
            if constexpr (PERT > 1) {
              const uint32_t i = tid + (PERT - 1) * BLOCK;
              if (i < ns_) rk[PERT - 1] = i + b1.lo; else if (i < n) rk[PERT - 1] = b1.lo + S.app_rank[i - ns_];
            }
          } else {
            // the near set was refilled from a far bucket (hundreds of unsorted entries, one batch in twenty): sort all
            // of it in place with a bitonic network, one compare-exchange per thread and stage.  Stages whose pairs
            // lie within 64 consecutive entries involve 32 consecutive threads -- one wave, whose LDS operations
            // execute in order -- so only the wider stages need a workgroup barrier (10 of the 55 for 1024 entries).
            // [ranking every entry against all others took ~150 k cycles here: 15 % of a deep query's time]
            static_assert(SM::NCAP <= 2 * BLOCK, "one compare-exchange per thread");
            uint32_t np2 = 512;
            while (np2 < n) np2 <<= 1;
            for (uint32_t i = n + tid; i < np2; i += BLOCK) {  // padding sorts last
              S.near_f[i] = INFINITY; S.near_g[i] = INFINITY; S.near_id[i] = 0xFFFFFFFFu; S.near_idx[i] = NIL;
            }
            lds_barrier();
            for (uint32_t k = 2; k <= np2; k <<= 1) {
              for (uint32_t jj = k >> 1; jj > 0; jj >>= 1) {
                const uint32_t t = (uint32_t)tid;
                if (t < (np2 >> 1)) {
                  const uint32_t a = ((t & ~(jj - 1u)) << 1) | (t & (jj - 1u)), b = a | jj;
                  const double fa = S.near_f[a], ga = S.near_g[a], fb = S.near_f[b], gb = S.near_g[b];
                  const uint32_t ia = S.near_id[a], ib = S.near_id[b];
                  const bool b_first = entry_less(fb, gb, ib, fa, ga, ia);
                  if (((a & k) == 0u) == b_first) {  // ascending run and b precedes a, or descending run and a precedes b
                    const uint32_t xa = S.near_idx[a], xb = S.near_idx[b];
                    S.near_f[a] = fb; S.near_g[a] = gb; S.near_id[a] = ib; S.near_idx[a] = xb;
                    S.near_f[b] = fa; S.near_g[b] = ga; S.near_id[b] = ia; S.near_idx[b] = xa;
                  }
                }
                const uint32_t next = jj > 1u ? (jj >> 1) : k;  // stride of the stage that follows
                if (jj > 32u || next > 32u) lds_barrier(); else unit_sync<64>();
              }
            }
            lds_barrier();
#pragma unroll
            for (int r = 0; r < PERT; r++) {
              const uint32_t i = tid + r * BLOCK;
              if (i < n) { ef[r] = S.near_f[i]; eg[r] = S.near_g[i]; ei[r] = S.near_id[i]; ex[r] = S.near_idx[i]; rk[r] = i; }
            }
          }
          MPLX_T2(S, 15, t3);
          lds_barrier();
          MPLX_T2(S, 18, t3);
#pragma unroll
          for (int r = 0; r < PERT; r++) {
            const uint32_t i = tid + r * BLOCK;
            if (i < n) {
              if (rk[r] < kc) {
                S.cand_f[rk[r]] = ef[r]; S.cand_g[rk[r]] = eg[r]; S.cand_id[rk[r]] = ei[r]; S.cand_idx[rk[r]] = ex[r];
              } else {
                const uint32_t pos = rk[r] - kc;
                S.near_f[pos] = ef[r]; S.near_g[pos] = eg[r]; S.near_id[pos] = ei[r]; S.near_idx[pos] = ex[r];
              }
            }
          }
          if (tid == 0) {
            S.n_cand = (int32_t)kc;
            S.n_near = n - kc;
            S.n_sorted = n - kc;
          }
          lds_barrier();
          MPLX_T2(S, 19, t3);
          if constexpr (HELP) {
            // wish list: the front of what is left of OPEN (sorted) = the candidates of the next batches.  Announced
            // by the seq store at the start of the NEXT batch, so no wait for these stores is needed here.
            if (S.helped && tid < WISH) {
              unsigned long long v = ~0ull;
              if ((uint32_t)tid < n - kc) v = ((unsigned long long)((uint32_t)q & 0xFFFFu) << 48) | (unsigned long long)Q.node_rec(S.near_id[tid]);
              st_u64(&(P.boxes + blockIdx.x)->wish[S.box_seq & 1ull][opaque(tid)], v);
              if (tid == 0) S.box_seq++;
            }
          }
        }
        const int n_cand = S.n_cand;
        // ---- 2a. fetch the candidates' records; drop stale entries (improved or closed since pushed)
        bool live_unit = false;
        [[maybe_unused]] unsigned long long hc_a = 0, hc_b = 0;
#if MPLX_X_EARLY_SETUP
        if (lu == 0) {  // the unit's words of the batch (a live unit sets cand_live below: same lane, program order)
          const int ko = opaque(ku);
          S.cand_live[ko] = 0;
          S.unit_seq[ko] = 0;
          S.cur_slot[ko] = NIL;
          S.u_goal[ko] = 0;
          S.u_cut[ko] = K;
          if constexpr (HELP) S.hc_row[ko] = 0;
        }
#endif
        if (ku < n_cand) {
          const uint32_t cid = S.cand_id[ku];
          double rg = pf_g, sval = pf_s;
          uint32_t fl = pf_fl;
          int32_t kval = pf_k;
          hc_a = pf_a; hc_b = pf_b;
          // [Two ways of sparing this fetch were measured and lost (round 3): handing the other units' prefetched records
          // over through LDS -- one fresh push among the first K shifts every later candidate by one unit -- costs more in
          // the selection (the stage write waits for the look-ahead cache words before its last barrier: +0.5 k cycles)
          // than the arrival skew it removes (-0.4 k); keeping the records of the states a batch creates in LDS for the
          // next batch's candidates: +1 %.]
          if (cid != pf_id) {  // (uniform per unit) not the prefetched entry: fetch it now
            char *rec = Q.node(cid);
            rg = V::g(rec);
            fl = V::flags(rec);
            if (lu <= ns + EX) sval = ld_state(&V::state(rec)[lu]);
            if (lu < NKY) kval = V::key(rec)[lu];
            if constexpr (HELP) {
              if (S.helped && lu == UL - 1) {
                const unsigned long long *cr = (const unsigned long long *)&P.cache_c[Q.node_rec(cid)];
                hc_a = ld_u64(cr);
                hc_b = ld_u64(cr + 1);
              }
            }
          }
          live_unit = __double_as_longlong(rg) == __double_as_longlong(S.cand_g[ku]) && !(fl & FLAG_CLOSED);
          if (live_unit) {
            const int ko = opaque(ku), lo = opaque(lu);
            if (lo < ns) S.cur[ko][lo] = sval;
            if (YAW && lo == ns) S.cur_yaw[ko] = sval;
            if (lo == ns + EX) S.cur[ko][12] = sval;
            if (lo >= ns && lo < 12) S.cur[ko][lo] = 0.0;
            if (lo < NKY) S.cur_key[ko][lo] = kval;
            if (lo == 0) {
              S.cand_live[ko] = 1;
              S.cand_fl[ko] = fl;
            }
          }
        }
        unit_sync<UL>();
        MPLX_T2(S, 20, t3);
        if constexpr (FILTER) {  // the candidate's key in the filter table: expandable?
          if (live_unit && lu == 0) {
            const FilterView F = filter_view(P);
            const unsigned long long hk = key_hash64(S.cur_key[ku], NKY);
            const unsigned long long tag = (hk >> 48) << 48;
            size_t pos = (size_t)hk & (size_t)F.mask;
            bool ok = false;
            for (unsigned long long steps = 0; steps <= F.mask; steps++) {
              const unsigned long long v = ld_u64(&F.table[pos]);
              if (v == TBL_EMPTY) break;
              const uint32_t vid = (uint32_t)v;
              if (vid < CLAIM_BASE && (v & 0xFFFFFFFF00000000ull) == tag) {
                const char *r = F.pool + (size_t)vid * rec_bytes(CONTROL);
                const int32_t *kk = (const int32_t *)(r + 24);
                uint32_t kd = 0;
#pragma unroll
                for (int i = 0; i < NKY; i++) kd |= (uint32_t)(kk[i] ^ S.cur_key[ku][i]);
                if (kd == 0u) {
                  ok = (*(const uint32_t *)(r + 16) & F.flag) != 0u;
                  break;
                }
              }
              pos = (pos + 1) & (size_t)F.mask;
            }
            if (!ok) S.cand_live[opaque(ku)] = 0;
          }
          unit_sync<UL>();
          live_unit = live_unit && S.cand_live[opaque(ku)] != 0;
        }
        if (live_unit && lu == 0) {  // goal test of the candidate (applied when, and if, it is committed)
          State sgoal;
          for (int i = 0; i < 12; i++) ((double *)&sgoal)[i] = S.cur[ku][i];
#if MPLX_X_LDS_CONST
          S.u_goal[ku] = (S.cur[ku][12] >= P.t_max || goal_reached(sgoal, YAW ? S.cur_yaw[ku] : 0.0)) ? 1 : 0;
#else
          S.u_goal[ku] = (S.cur[ku][12] >= P.t_max || goal_reached(sgoal, YAW ? S.cur_yaw[ku] : 0.0)) ? 1 : 0;
#endif
        }
        if constexpr (HELP) {
          // did a helper expand this node ahead of time?  The entry must be of THIS state: the helper's hash of the
          // key of the state it expanded against the candidate's own key (guards against anything stale on its side)
          if (live_unit && S.helped && !MPLX_XF(P, 256) && lu == UL - 1 && (uint32_t)hc_a != 0u && ((uint32_t)hc_b & CACHE_READY) &&  // [MPLX_X_FLAGS & 256, diagnostics: no hit is taken]
              (uint32_t)(hc_a >> 32) == (uint32_t)key_hash64(S.cur_key[ku], nk)) {
            S.hc_row[ku] = (uint32_t)hc_a;
            S.hc_valid[ku] = (uint32_t)hc_b;
            S.hc_blocked[ku] = (uint32_t)(hc_b >> 32);
            atomicAdd(&S.cyc[9], 1ull);  // (diagnostics) candidates served from the look-ahead cache
          }
        }
        unit_sync<UL>();
        // the helper's row of a cached candidate -- the heuristics of its successors, the voxel-read count -- is asked
        // for now (agent-scope loads: always a trip to memory) so that it travels during the expansion; used in 2b / 2c
        [[maybe_unused]] double h_row = 0.0;
        [[maybe_unused]] unsigned long long reads_word = 0ull;  // (units of 32 lanes: voxel reads | check word of the row << 32)
        [[maybe_unused]] unsigned long long h_tag = 0ull, reads_tag = 0ull;  // (MPLX_X_ROW_PAIRS: the checks of this lane's heuristic and of the reads)
        [[maybe_unused]] uint32_t reads_row = 0u;  // (32 bits asked for: a 64-bit load whose upper half is dead makes the compiler wait for it at once, to reuse the register)
#if MPLX_X_EARLY_ROW
        if constexpr (HELP) {
          const uint32_t rp1 = live_unit ? S.hc_row[ku] : 0u;
          if MPLX_XF(P, 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          if (rp1) {
            const double *row = P.cache_h + (size_t)(rp1 - 1u) * cache_row_doubles(UL);
            bool want = lu < P.n_u && P.eps != 0.0;
            if constexpr (UL <= 64) want = want && (((S.hc_valid[ku] & ~S.hc_blocked[ku]) >> lu) & 1u);
            if (want) h_row = ld_f64_agent(&row[cache_h_slot(UL, lu)]);
#if MPLX_X_ROW_PAIRS
            if constexpr (UL == 32) {
              if (want) h_tag = ld_u64((const unsigned long long *)&row[cache_h_slot(UL, lu) + 1]);
              if (lu == 0) reads_tag = ld_u64((const unsigned long long *)&row[cache_reads_slot(UL) + 1]);
            }
#endif
            if (lu == 0) {
              if constexpr (UL == 32) {
                reads_word = ld_u64((const unsigned long long *)&row[cache_reads_slot(UL)]);
                reads_row = (uint32_t)reads_word;
              } else {
                reads_row = ld_u32((const uint32_t *)&row[cache_reads_slot(UL)]);
              }
            }
          }
        }
#endif
        MPLX_T2(S, 21, t3);
        MPLX_TOC(S, 0, tp);
        // ---- 2b. expand all live units concurrently
        MPLX_TIC(tx);
        LaneSucc L;
        // first probe of the state-space table: issued as soon as the successor's key exists, i.e. before
        // its voxels are sampled (a blocked successor wastes one load), consumed after the batch table is built
        unsigned long long h64 = 0, v0 = TBL_EMPTY;
        [[maybe_unused]] unsigned long long v1 = TBL_EMPTY;  // (MPLX_X_PROBE2) the slot behind the home slot, loaded with it
        size_t pos0 = 0;
        expand_unit<UL, BLOCK, CONTROL, HELP, POT, YAW>(P, S, tid, live_unit, L, [&](const LaneSucc &l) {
          if (l.valid && !l.blocked) {
            h64 = lane_hash(l);
            pos0 = (size_t)(h64 ^ ((unsigned long long)(uint32_t)q * 0x9E3779B97F4A7C15ull)) & (size_t)P.table_mask;
            v0 = MPLX_XF(P, 1) ? ld_u64(&P.table[pos0]) : ld_u64_probe(&P.table[pos0]);
#if MPLX_X_PROBE2
            v1 = ld_u64_probe(&P.table[(pos0 + 1) & (size_t)P.table_mask]);
#endif
          }
        });
        const bool act = L.valid && !L.blocked;
        {
          uint32_t tot, treads;
          unit_excl_scan<UL, BLOCK>((L.valid ? 1u : 0u) | (act ? 1u << 10 : 0u), S, tid, tot);
          unit_excl_scan<UL, BLOCK>(L.reads, S, tid, treads);
          if (lu == 0) {
            S.u_succ[ku] = tot & 0x3FFu;
            S.u_fin[ku] = tot >> 10;
            if (HELP && S.hc_row[ku] != 0u)  // voxel reads of the expansion as the helper counted them (slot 31 of its row)
#if MPLX_X_EARLY_ROW
              S.u_reads[ku] = reads_row;  // (units of 32 lanes: written again once the row has passed its check, before the barrier of 2c)
#else
              S.u_reads[ku] = (uint32_t)ld_u64((const unsigned long long *)&P.cache_h[(size_t)(S.hc_row[ku] - 1u) * cache_row_doubles(UL) + cache_reads_slot(UL)]);
#endif
            else
              S.u_reads[ku] = treads;
          }
        }
        MPLX_TOC(S, 1, tx);
        // ---- 2c. batch table: one entry per distinct successor key; its leader looks the key up in
        //          the state space (or claims a slot) and computes the heuristic of a new state
        MPLX_TIC(tc);
        [[maybe_unused]] unsigned long long t2 = __builtin_readcyclecounter();
        int my_slot = 0;
        bool claimed_new = false;  // this lane claimed a slot of the state table for a state that does not exist yet ...
        size_t claimed_pos = 0;    // ... this one: it holds the state's entry, or TBL_DEAD_ID, before the batch ends
        // (the table was cleared by the idle waves of the previous batch's bookkeeping -- batch_setup -- unless the
        // expansion borrowed a column for the potential sums)
        constexpr bool CLEAR_HERE = !(MPLX_X_EARLY_SETUP && MPLX_X_EARLY_CLEAR) || POT;
        if constexpr (CLEAR_HERE) {
          for (int i = tid; i < BT; i += BLOCK) {
            S.bt_hash[i] = 0ull;
            S.bt_leader[i] = NIL;
            S.bt_dirty[i] = 0;
            S.bt_share[i] = 0;
          }
        }
        if (act) {
#pragma unroll
          for (int i = 0; i < nk; i++) S.lane_key[tid][i] = L.key[i];
          if constexpr (YAW) S.lane_key[tid][nk] = L.yaw_key;
        }
        MPLX_T2(S, 14, t2);
#ifdef MPLX_LOOKUP_TIMERS
        if ((tid & 63) == 0) S.cycw[tid >> 6][0] += __builtin_readcyclecounter() - tx;
        unsigned long long tw = 0;
#endif
        if constexpr (CLEAR_HERE) __syncthreads();
        MPLX_T2(S, 0, t2);
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
          // two lanes of ONE unit on one state (duplicate control inputs, or two inputs quantising to one
          // key): that unit commits lane by lane when the batch takes the ordered path.  Bits 8.. of
          // bt_dirty collect the units that reach the entry (bit 0 is the ordered path's dirty flag).
          const uint32_t ubit = 0x100u << ku;
          if (atomicOr(&S.bt_dirty[sl], ubit) & ubit) S.unit_seq[ku] = 1;
        }
#if MPLX_X_EARLY_ROW && !MPLX_X_ROW_PAIRS
        if constexpr (HELP && UL == 32) {
          // the helper's row against its check word (cache_row_term above): what does not match yet is a store on its way.
          // (Placed here, not right after the expansion: the row's agent-scope loads keep travelling during the scans and the
          //  batch-table insert above.)
          const uint32_t rp1 = live_unit ? S.hc_row[ku] : 0u;  // (the same for the 32 lanes of the unit)
          if (rp1 && !MPLX_ROW_FENCE(P) && !MPLX_XF(P, 1024)) {  // [MPLX_X_FLAGS & 1024, measurement: the row is not checked (round 3's protocol)]
            const double *row = P.cache_h + (size_t)(rp1 - 1u) * cache_row_doubles(UL);
            const bool want = act && P.eps != 0.0;  // (the lanes whose heuristic was asked for: the record's masks are L.valid / L.blocked now)
            const uint32_t khash = (uint32_t)key_hash64(S.cur_key[ku], nk);
            for (uint32_t polls = 0;; polls++) {
              const uint32_t cs = unit32_xor(want ? cache_row_term(h_row, lu) : 0u);
              const uint32_t rd = (uint32_t)__shfl((int)(uint32_t)reads_word, 0, 32), stored = (uint32_t)__shfl((int)(uint32_t)(reads_word >> 32), 0, 32);
              if ((cs ^ cache_row_salt(khash, (uint32_t)q, P.epoch, rd)) == stored) break;
              if (polls >= CACHE_ROW_POLLS) {  // (never seen: a record whose row did not arrive)
                if (lu == 0) S.status = 5;
                break;
              }
              if ((polls & (GUARD_POLL_EVERY - 1u)) == GUARD_POLL_EVERY - 1u && lu == 0) guard_mark(P, GUARD_ROW_WAIT, (uint32_t)q, S.cyc[7], (unsigned long long)rp1);
              __builtin_amdgcn_s_sleep(16);
              if (want) h_row = ld_f64_agent(&row[cache_h_slot(UL, lu)]);
              if (lu == 0) {
                reads_word = ld_u64((const unsigned long long *)&row[cache_reads_slot(UL)]);
                reads_row = (uint32_t)reads_word;
              }
            }
            if (lu == 0) S.u_reads[ku] = reads_row;
          }
        }
#endif
        __syncthreads();
        MPLX_T2(S, 1, t2);
#ifdef MPLX_LOOKUP_TIMERS
        tw = __builtin_readcyclecounter();
#endif
        double hspec = 0.0;
        if (act) {
          const uint32_t leader = S.bt_leader[my_slot];
          if (leader != (uint32_t)tid) {
            uint32_t kd = 0;
#pragma unroll
            for (int i = 0; i < nk; i++) kd |= (uint32_t)(S.lane_key[leader][i] ^ L.key[i]);
            if constexpr (YAW) kd |= (uint32_t)(S.lane_key[leader][nk] ^ L.yaw_key);
            if (kd != 0u) S.status = 5;                                    // 64-bit key-hash collision inside a batch
            // a state reached from two lanes of the batch: harmless while both only append a predecessor
            // edge (decided once g is known); three lanes on one state take the ordered path
            if (atomicAdd(&S.bt_share[my_slot], 0x10000u + (uint32_t)tid) != 0u) S.batch_dep = 1;
            S.any_shared = 1;
#ifdef MPLX_DEP_STATS
            atomicOr(&S.dep_cause, 1);
#endif
          } else {
            const unsigned long long tagq = tbl_tagq(h64, (uint32_t)q, P.tbl_epoch);
            const size_t mask = (size_t)P.table_mask;
            size_t pos = pos0;
            const uint32_t claim_batch = (batch_no & CLAIM_BATCH_MASK) << CLAIM_BATCH_SHIFT;
            static_assert(BLOCK <= (1 << CLAIM_BATCH_SHIFT), "the thread index of a claim has nine bits");
            const unsigned long long claim = tagq | (unsigned long long)(CLAIM_BASE + claim_batch + (uint32_t)tid);
            // second round trip (claim the empty slot, or fetch the record the slot names) goes out
            // before the heuristic is computed, so its latency hides behind the f64 work.  (Computing the
            // heuristic only for states found to be new, after the look-up, was measured slower: the slowest
            // lane of the workgroup sets the pace either way, and the overlap is lost.)
            unsigned long long cas0 = 0;
            bool did_cas0 = false;
            if (tbl_empty(v0, P.tbl_epoch)) {  // (cleared, or left by a batch of another epoch: claimed against the value seen)
              cas0 = atomicCAS(&P.table[pos], v0, claim);
              did_cas0 = true;
            } else if ((uint32_t)v0 < CLAIM_BASE && (v0 & 0xFFFFFFFF00000000ull) == tagq) {
              __builtin_prefetch(Q.node((uint32_t)v0), 0, 3);
            }
            MPLX_T2(S, 2, t2);
            if (P.eps != 0.0) {
              // look-ahead cache hit: the heuristic of this successor is in the helper's row (written before the
              // cache record was, read after it)
              if (HELP && S.hc_row[ku] != 0u)
#if MPLX_X_EARLY_ROW
              {
#if MPLX_X_ROW_PAIRS
                if constexpr (UL == 32) {  // this lane's {heuristic, check}: what does not match yet is a store on its way
                  const uint32_t kh = (uint32_t)key_hash64(S.cur_key[ku], nk);
                  const double *row = P.cache_h + (size_t)(S.hc_row[ku] - 1u) * cache_row_doubles(UL);
                  for (uint32_t polls = 0; cache_pair_tag((unsigned long long)__double_as_longlong(h_row), kh, (uint32_t)q, P.epoch, (uint32_t)lu) != h_tag; polls++) {
                    if (polls >= CACHE_ROW_POLLS) { S.status = 5; break; }
                    __builtin_amdgcn_s_sleep(16);
                    h_row = ld_f64_agent(&row[cache_h_slot(UL, lu)]);
                    h_tag = ld_u64((const unsigned long long *)&row[cache_h_slot(UL, lu) + 1]);
                  }
                }
#endif
                hspec = h_row;
              }
#else
                hspec = ld_f64_agent(&P.cache_h[(size_t)(S.hc_row[ku] - 1u) * cache_row_doubles(UL) + cache_h_slot(UL, lu)]);
#endif
              else  // (get_heur is 0 when the state's key equals the goal's: with yaw the yaw key is part of that comparison)
                hspec = (YAW && L.yaw_key != S.hp.goal_yaw_key) ? cal_heur(S.hp, CONTROL, L.tn) : get_heur(S.hp, CONTROL, L.tn, L.key, nk);
            }
            MPLX_T2(S, 3, t2);
#ifdef MPLX_LOOKUP_TIMERS
            atomicMax(&S.arr_heur, (unsigned long long)__builtin_readcyclecounter());
#endif
            bool first = true;
            for (uint32_t steps = 0;; steps++) {
              if (steps > (1u << 22)) {  // (a probe never walks this far in a table four times the node capacity)
                S.status = 5;
                guard_mark(P, GUARD_PROBE, (uint32_t)q, S.cyc[7], (unsigned long long)pos);
                break;
              }
#if MPLX_X_PROBE2
              // (a view of the second slot as old as the first one's: rule R3 covers it the same way -- a stale EMPTY is caught by the
              //  compare-and-swap, an own-tag claim is read again past the L1, anything foreign moves the probe on)
              unsigned long long v = first ? v0 : (steps == 1u && !MPLX_XF(P, 1)) ? v1 : MPLX_XF(P, 1) ? ld_u64(&P.table[pos]) : ld_u64_probe(&P.table[pos]);
#else
              unsigned long long v = first ? v0 : MPLX_XF(P, 1) ? ld_u64(&P.table[pos]) : ld_u64_probe(&P.table[pos]);
#endif
              if (tbl_empty(v, P.tbl_epoch)) {
                unsigned long long old = (first && did_cas0) ? cas0 : atomicCAS(&P.table[pos], v, claim);
                first = false;
                if (old == v) {
                  S.bt_id[my_slot] = NIL;  // new state; created when its first sharer commits
                  S.bt_tslot[my_slot] = (uint32_t)pos;
                  claimed_new = true;
                  claimed_pos = pos;
                  S.bt_h[my_slot] = hspec;  // of the leader's state = the state the node will be created with
                  S.bt_g[my_slot] = INFINITY;
                  S.bt_flags[my_slot] = 0;
                  S.bt_pred[my_slot] = NIL;
                  break;
                }
                v = old;
              }
              first = false;
              // A claim with this query's tag seen through the compute unit's L1 may be a STALE copy of a slot that has
              // long held its final entry (the line was fetched between the claiming compare-and-swap and the agent-scope
              // store of the entry, which does not update the L1 copy): taken at face value the probe would move on and
              // create the state a second time.  Look again past the L1 before believing it.
              // And a claim of an EARLIER batch that memory still shows is a store on its way: the lane that made it wrote the
              // state's entry, or TBL_DEAD_ID, before that batch ended (below, after the commit) -- an agent-scope store that is
              // posted, not awaited, and that a loaded memory system (another launch's hipMemset, a neighbour's copy kernel) can
              // hold back for longer than a batch lasts.  Moving on here would create the state a second time (seen: blocking
              // batches under a background fill load, streamed batches -- profiles/r04s_*); wait for the store instead.
              if ((uint32_t)v >= CLAIM_BASE && (v & 0xFFFFFFFF00000000ull) == tagq) {
                v = ld_u64(&P.table[pos]);
                for (uint32_t polls = 0; !MPLX_XF(P, 2048) /* [measurement: no wait, round 3's rule] */ && (uint32_t)v >= CLAIM_BASE && (uint32_t)v < TBL_DEAD_ID && (v & 0xFFFFFFFF00000000ull) == tagq &&
                                         ((uint32_t)v & (CLAIM_BATCH_MASK << CLAIM_BATCH_SHIFT)) != claim_batch; polls++) {
                  if (polls >= CLAIM_WAIT_POLLS) { S.status = 5; break; }  // (never seen: a claim nobody resolved)
                  if ((polls & (GUARD_POLL_EVERY - 1u)) == GUARD_POLL_EVERY - 1u) guard_mark(P, GUARD_CLAIM_WAIT, (uint32_t)q, S.cyc[7], (unsigned long long)pos);
                  __builtin_amdgcn_s_sleep(16);
                  v = ld_u64(&P.table[pos]);
                }
              }
              const uint32_t vid = (uint32_t)v;
              if (vid < CLAIM_BASE && (v & 0xFFFFFFFF00000000ull) == tagq) {
                // the record's hot line in 16-byte words, all asked for at once: g | h | flags pred key[0..1] | key[2..] ...
                const uint4 *r4 = (const uint4 *)Q.node(vid);
                constexpr int NW4 = (24 + 4 * NKY + 15) / 16;
                uint32_t w[4 * NW4];
#pragma unroll
                for (int j = 0; j < NW4; j++) {
                  const uint4 x = r4[j];
                  w[4 * j] = x.x; w[4 * j + 1] = x.y; w[4 * j + 2] = x.z; w[4 * j + 3] = x.w;
                }
                uint32_t kd = 0;
#pragma unroll
                for (int i = 0; i < nk; i++) kd |= w[6 + i] ^ (uint32_t)L.key[i];
                if constexpr (YAW) kd |= w[6 + nk] ^ (uint32_t)L.yaw_key;
                const double rg = __hiloint2double((int)w[1], (int)w[0]), rh = __hiloint2double((int)w[3], (int)w[2]);
                const uint32_t rfl = w[4], rpred = w[5];
                if (kd == 0u) {
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
#if MPLX_X_EARLY_ROW && MPLX_X_ROW_PAIRS
        if constexpr (HELP && UL == 32) {  // the helper's voxel-read count of a cached unit, once its pair checks out
          if (lu == 0 && live_unit && S.hc_row[ku] != 0u) {
            const uint32_t kh = (uint32_t)key_hash64(S.cur_key[ku], nk);
            const double *row = P.cache_h + (size_t)(S.hc_row[ku] - 1u) * cache_row_doubles(UL);
            for (uint32_t polls = 0; cache_pair_tag(reads_word, kh, (uint32_t)q, P.epoch, 63u) != reads_tag; polls++) {
              if (polls >= CACHE_ROW_POLLS) { S.status = 5; break; }
              __builtin_amdgcn_s_sleep(16);
              reads_word = ld_u64((const unsigned long long *)&row[cache_reads_slot(UL)]);
              reads_tag = ld_u64((const unsigned long long *)&row[cache_reads_slot(UL) + 1]);
            }
            S.u_reads[ku] = (uint32_t)reads_word;
          }
        }
#endif
        MPLX_T2(S, 4, t2);
#ifdef MPLX_LOOKUP_TIMERS
        if ((tid & 63) == 0) S.cycw[tid >> 6][1] += __builtin_readcyclecounter() - tw;
#endif
        // does a candidate itself appear among the successors of the batch?  (its closed flag must
        // reach the units committed after it)
        if (lu == 0 && live_unit) {
          const unsigned long long hk = key_hash64(S.cur_key[ku], NKY);
          const unsigned long long hv = hk | 1ull;
          int sl = (int)(hk >> 7) & (BT - 1);
          for (;;) {
            const unsigned long long o = S.bt_hash[sl];
            if (o == 0ull) break;
            if (o == hv) {
              S.cur_slot[ku] = (uint32_t)sl;
              atomicOr(&S.bt_share[sl], 0x80000000u);  // its closed flag matters to a lane that would re-open it
#ifdef MPLX_DEP_STATS
              atomicOr(&S.dep_cause, 2);
#endif
              break;
            }
            sl = (sl + 1) & (BT - 1);
          }
        }
        MPLX_T2(S, 22, t2);
#ifdef MPLX_LOOKUP_TIMERS
        if ((tid & 63) == 0) S.cycw[tid >> 6][2] += __builtin_readcyclecounter() - tw;
        atomicMax(&S.arr_max, (unsigned long long)__builtin_readcyclecounter());
#endif
        __syncthreads();
#ifdef MPLX_LOOKUP_TIMERS
        if ((tid & 63) == 0) S.cycw[tid >> 6][3] += __builtin_readcyclecounter() - tw;
        if (tid == 0) { S.cyc2[23] += __builtin_readcyclecounter() - S.arr_max; S.sum_heur += S.arr_heur - tw; S.sum_arr += S.arr_max - tw; }
#endif
        MPLX_T2(S, 5, t2);
        LanePre pre;
        pre.tg = 0.0; pre.pf = INFINITY; pre.code = -1; pre.cut = K;
        if (act) {
          const bool nw = S.bt_id[my_slot] == NIL;
          const double hv = S.bt_h[my_slot];
#if MPLX_X_LDS_CONST
          const double uc = S.ucost_lds[lu];
#else
          const double uc = P.ucost[lu];
#endif
          pre.tg = S.cand_g[ku] + ((POT && P.map.aux) ? uc + P.pot_weight * (double)L.pot : uc);
          pre.pf = pre.tg + P.eps * hv;
          if (pre.pf != pre.pf) pre.pf = INFINITY;
          // a state created by this batch gets an id above every existing one: ties on (f, g) never favour it
          const uint32_t pid = nw ? 0xFFFFFFFFu : S.bt_id[my_slot];
          pre.code = classify(S, P.bucket_width, pre.pf, pre.tg, pid);
          // candidates are in ascending order: "my entry precedes candidate k2" is monotone in k2; most entries precede
          // none (one look at the last candidate), and a wave whose lanes all see that skips the search
          if (ku + 1 < n_cand && entry_less(pre.pf, pre.tg, pid, S.cand_f[n_cand - 1], S.cand_g[n_cand - 1], S.cand_id[n_cand - 1])) {
            int lo = ku + 1, hi = n_cand - 1;  // (precedes the last one: the answer is in [ku + 1, n_cand - 1])
            while (lo < hi) {
              const int mid = (lo + hi) >> 1;
              if (entry_less(pre.pf, pre.tg, pid, S.cand_f[mid], S.cand_g[mid], S.cand_id[mid])) hi = mid; else lo = mid + 1;
            }
            if (lo < n_cand) pre.cut = lo;
          }
          // valid when no state of the batch is shared between lanes (bt_g still is the fetched value)
          if (pre.cut < K && pre.tg < S.bt_g[my_slot]) atomicMin(&S.u_cut[ku], pre.cut);
          // a shared state (or a candidate's own state) that this lane creates or improves: units interact
          if (S.bt_share[my_slot] != 0u && (nw || pre.tg < S.bt_g[my_slot])) S.batch_dep = 1;
#ifdef MPLX_DEP_STATS
          if (pre.tg < S.bt_g[my_slot]) atomicOr(&S.bt_dirty[my_slot], 2u);  // some sharer modifies this state
#endif
        }
        MPLX_T2(S, 6, t2);
        MPLX_TOC(S, 2, tc);
        if (S.status >= 0) break;
        // ---- 3. ordered commit
        MPLX_TIC(to);
        lds_barrier();  // the per-unit cut points (atomicMin above) are complete
#ifdef MPLX_DEP_STATS
        if (act && S.bt_leader[my_slot] != (uint32_t)tid && (S.bt_dirty[my_slot] & 2u)) atomicOr(&S.dep_cause, 4);
        lds_barrier();
        if (tid == 0) {
          const int dc = S.dep_cause;
          if (dc & 1) S.cyc[3]++;
          if (dc & 2) S.cyc[5]++;
          if (dc & 4) S.cyc[9]++;
        }
        for (int i = tid; i < BT; i += BLOCK) S.bt_dirty[i] = 0;
        lds_barrier();
#endif
        int k_stop = n_cand, n_commit = 0;
        const unsigned long long expanded0 = S.c_expanded;
        const bool parallel_commit = !S.batch_dep;
        MPLX_T2(S, 7, t2);
        if (parallel_commit) {
          // no state is touched by two lanes of the batch and no candidate is a successor inside it:
          // the units cannot see each other, so which of them commit follows from the per-unit cut
          // points alone, and they commit together
          // (every wave evaluates this redundantly: lane k holds unit k, ballots make the result uniform)
          int st_after = -1;
          {
            const int l = opaque(tid & 63);
            const bool inb = l < K && l < n_cand;
            const bool lvk = inb && S.cand_live[inb ? l : 0] != 0;
            const int uck = lvk ? S.u_cut[l] : K;
            const bool ugk = lvk && S.u_goal[l] != 0;
            const int c = row_incl_min(uck, K);  // inclusive prefix minimum of the cut points of the live units
            const int cex = (int)dpp_u32<0x111, 0xf>((uint32_t)K, (uint32_t)c);  // cut point set by the units before mine (lane 0: none)
            const unsigned long long livem = __ballot(lvk);
            const unsigned long long le = ~0ull >> (63 - l);
            const int cnt = __popcll(livem & le);  // live units up to and including mine
            const bool stop_a = lvk && l >= cex;   // an earlier unit's entry precedes this candidate
            const bool ok = lvk && l < cex;
            const bool stop_b = ok && ugk;          // goal reached when this unit is committed
            const bool stop_c = ok && !ugk && P.max_expand > 0 && expanded0 + (unsigned long long)cnt >= (unsigned long long)P.max_expand;
            const unsigned long long ma = __ballot(stop_a), mb = __ballot(stop_b), mc = __ballot(stop_c);
            const unsigned long long many = ma | mb | mc;
            if (many) {
              const int first = __ffsll((long long)many) - 1;
              const unsigned long long bit = 1ull << first;
              if (ma & bit) {
                k_stop = first;
                n_commit = __popcll(livem & (bit - 1ull));
              } else {
                k_stop = first + 1;
                n_commit = __popcll(livem & (bit | (bit - 1ull)));
                st_after = (mb & bit) ? 0 : 3;
              }
            } else {
              n_commit = __popcll(livem);
            }
          }
          const bool mine = ku < k_stop && S.cand_live[opaque(ku)];
          if (mine && lu == 0) V::flags(Q.node(S.cand_id[ku])) = S.cand_fl[ku] | FLAG_CLOSED;
          if (MPLX_EARLY_TOMB(P)) {
            if (act && mine) atomicOr(&S.bt_dirty[my_slot], 2u);  // (bit 1, free in the parallel commit: a committed unit reaches this state)
          }
          MPLX_T2(S, 8, t2);
          __syncthreads();  // everyone has read status / u_cut before they change
          MPLX_T2(S, 9, t2);
          // a slot claimed for a state no committed unit reaches: its TBL_DEAD_ID goes out with the commit's own stores (behind the
          // commit's barrier its write-through acknowledgement would be the first thing the next batch waits for)
          if (MPLX_EARLY_TOMB(P) && !MPLX_XF(P, 4096) && claimed_new && !(S.bt_dirty[my_slot] & 2u)) {
            st_u64(&P.table[claimed_pos], tbl_tagq(h64, (uint32_t)q, P.tbl_epoch) | (unsigned long long)TBL_DEAD_ID);
            claimed_new = false;
          }
          spec_commit_lanes<UL, K, CONTROL, true, HELP, YAW>(Q, S, tid, q, ku, act && mine, my_slot, k_stop, L, hspec, pre, pend_idx, pend_old, (int)(batch_no & 1u));
          if (tid == 0 && st_after >= 0) S.status = st_after;
          MPLX_T2(S, 10, t2);
        }
        for (int k = 0; k < (parallel_commit ? 0 : n_cand); k++) {
          if (!S.cand_live[k]) continue;  // stale entry: dropped, like a pop that skips it
          if (k >= S.cut_at) {            // something pushed by this batch now precedes candidate k
            k_stop = k;
            break;
          }
          n_commit++;
          if (tid == k * UL) {
            const uint32_t cur = S.cand_id[k];
            // flags cannot have changed since they were fetched: any unit of this batch that improved
            // this node pushed an entry that precedes it and cut the batch before it
            if (S.cur_slot[k] != NIL) {  // also a successor of this batch: the batch table owns its flags
              S.bt_flags[S.cur_slot[k]] |= FLAG_CLOSED;
              S.bt_dirty[S.cur_slot[k]] = 1;
            } else {
              V::flags(Q.node(cur)) = S.cand_fl[k] | FLAG_CLOSED;
            }
            if (S.u_goal[k])
              S.status = 0;
            else if (P.max_expand > 0 && expanded0 + (unsigned long long)n_commit >= (unsigned long long)P.max_expand)
              S.status = 3;
          }
          if (S.cur_slot[k] != NIL) lds_barrier();  // (uniform) its closed flag must precede its own successors' relax
          if (!S.unit_seq[k]) {
            spec_commit_lanes<UL, K, CONTROL, false, HELP, YAW>(Q, S, tid, q, k, act && ku == k, my_slot, k_stop, L, hspec, pre, pend_idx, pend_old);
          } else {
            for (int i = 0; i < P.n_u; i++) {
              spec_commit_lanes<UL, K, CONTROL, false, HELP, YAW>(Q, S, tid, q, k, act && ku == k && lu == i, my_slot, k_stop, L, hspec, pre, pend_idx, pend_old);
              lds_barrier();
            }
          }
          lds_barrier();
          if (S.status >= 0) { k_stop = k + 1; break; }
        }
        MPLX_T2(S, 11, t2);
        if (!parallel_commit) {
          lds_barrier();
          // write the batch table back: one store per field and state, whatever number of units touched it
          for (int i = tid; i < BT; i += BLOCK) {
            if ((S.bt_dirty[i] & 1u) && S.bt_id[i] != NIL) {
              char *rec = Q.node(S.bt_id[i]);
              V::g(rec) = S.bt_g[i];
              V::flags(rec) = S.bt_flags[i];
              V::pred(rec) = S.bt_pred[i];
            }
          }
        }
        __syncthreads();
        MPLX_T2(S, 12, t2);
        // a slot claimed for a state that no committed unit reached (its units were cut): dead from here on, and said so -- no claim
        // outlives its batch (see the look-up above)
        if (claimed_new && S.bt_id[my_slot] == NIL && !MPLX_XF(P, 4096))  // [MPLX_X_FLAGS & 4096, measurement: abandoned claims stay claims (round 3)]
          st_u64(&P.table[claimed_pos], tbl_tagq(h64, (uint32_t)q, P.tbl_epoch) | (unsigned long long)TBL_DEAD_ID);
        if (tid < 64) {  // counters of the committed units, in commit order; lane k holds unit k
          const int l = opaque(tid);
          const bool inb = l < K && l < n_cand && S.cand_live[l < K ? l : 0] != 0;
          const bool done = inb && l < k_stop;
          const bool back = inb && l >= k_stop && S.status < 0;  // behind a cut: returns to OPEN untouched
          const uint32_t cur = l < K ? S.cand_id[l] : 0u;
          const unsigned long long m = __ballot(done), mb = __ballot(back), live_m = __ballot(inb);
          const unsigned long long below = (1ull << l) - 1ull;
          const unsigned long long ne0 = S.c_expanded;
          // (lanes >= K hold 0: the sums over the first row of 16 lanes are the totals)
          const uint32_t su = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_sum<16>(done ? S.u_succ[l] : 0u), 15);
          const uint32_t fi = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_sum<16>(done ? S.u_fin[l] : 0u), 15);
          const uint32_t rd = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_sum<16>(done ? S.u_reads[l] : 0u), 15);
          if (done && P.rec_ids) {
            const unsigned long long at = ne0 + (unsigned long long)__popcll(m & below);
            if (at < P.cap_rec) P.rec_ids[(size_t)q * P.cap_rec + at] = (int32_t)cur;
          }
          if (back) {
            const uint32_t pos = S.n_near + (uint32_t)__popcll(mb & below);
            S.near_f[pos] = S.cand_f[l]; S.near_g[pos] = S.cand_g[l]; S.near_id[pos] = cur; S.near_idx[pos] = S.cand_idx[l];
          }
          // running hash of the expansion order, h <- h P + (id + 1) per committed unit in order (mod 2^64), unrolled
          // algebraically: h P^n + sum_k (id_k + 1) P^(number of committed units after k); each lane forms its own
          // term (powers from a table in LDS), a row scan adds them up
          static_assert(K <= 16, "exponents up to 16: S.hpow");
          const unsigned long long term = row_incl_sum64(done ? (unsigned long long)(cur + 1u) * S.hpow[__popcll((m >> l) >> 1) & 31] : 0ull);
          const unsigned long long terms = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(term >> 32), 15) << 32) |
                                           (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)term, 15);
          const unsigned long long hh = S.c_hash * S.hpow[__popcll(m) & 31] + terms;
          if (m && l == 63 - __clzll((long long)m)) S.cur_id = cur;
          if (l == 0) {
            const unsigned long long ncm = (unsigned long long)__popcll(m);
            S.c_expanded = ne0 + ncm;
            S.c_closed += ncm;
            S.c_hash = hh;
            S.c_prims += ncm * (unsigned long long)P.n_u;
            S.c_succ += su;
            S.c_succ_finite += fi;
            S.c_reads += rd;
            S.n_near += (uint32_t)__popcll(mb);
            // speculation accounting: candidates taken / live units (they ran get_succ) / of those, cut and returned to OPEN.  (a batch
            // that ends the query leaves its uncommitted live units uncounted as cut: nothing expands them again)
            S.c_cand += (unsigned long long)n_cand;
            S.c_live += (unsigned long long)__popcll(live_m);
            S.c_cut += (unsigned long long)__popcll(mb);
            if (!parallel_commit) S.cyc[8]++;  // batches that needed the unit-by-unit commit
          }
        }
#if MPLX_X_EARLY_SETUP
        batch_setup(S.c_expanded);  // (thread 0 reads back what it has just written; nobody else uses the argument)
#endif
        __syncthreads();
        MPLX_T2(S, 13, t2);
        MPLX_TOC(S, 6, to);
        if (S.status >= 0) break;
      }
    }
    const uint32_t goal_id = searched ? S.cur_id : NIL;
    if (searched) clear_buckets(Q, tid);
    if constexpr (HELP) {
      if (tid == 0) st_u64(&(P.boxes + blockIdx.x)->seq, (unsigned long long)P.epoch << 32);  // helpers of this query detach
    }
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
        bool ok = true, too_long = false;
        while (V::pred(Q.node(node)) != NIL) {
          uint32_t best = NIL;
          double min_rhs = INFINITY, min_g = INFINITY;
          uint32_t hops = 0;
          for (uint32_t e = V::pred(Q.node(node)); e != NIL && hops <= S.n_edges; e = Q.edge(e)->next, hops++) {
            const EdgeRec er = *Q.edge(e);
            double gp = V::g(Q.node(er.parent));
            const double ec = (POT && P.map.aux) ? P.ucost[er.action & EDGE_ACTION_MASK] + P.pot_weight * (double)(er.action >> EDGE_POT_SHIFT) : P.ucost[er.action & EDGE_ACTION_MASK];
            double rhs = gp + ec;
            if (rhs < min_rhs || (rhs == min_rhs && gp >= min_g)) { min_rhs = rhs; min_g = gp; best = e; }
          }
          if (best == NIL) { ok = false; break; }
          if (len >= MAX_TRAJ) { too_long = true; break; }
          ta[len] = (int32_t)(Q.edge(best)->action & EDGE_ACTION_MASK);
          node = Q.edge(best)->parent;
          len++;
          tn[len] = (int32_t)node;
          if (node == 0u) break;
        }
        if (too_long) {  // goal reached and cost known; the path does not fit the device-side buffer
          cost = V::g(Q.node(goal_id));
          status = 6;    // MPLX_PLAN_TRAJ_TOO_LONG
          len = 0;
        } else if (ok) {
          cost = V::g(Q.node(goal_id));
          for (int i = 0; i <= len; i++) {
            const double *st = V::state(Q.node((uint32_t)tn[i]));
            for (int k = 0; k < 12; k++) ts[i * 13 + k] = k < ns ? ld_state(&st[k]) : 0.0;
            ts[i * 13 + 12] = ld_state(&st[ns + EX]);
            if (YAW && P.traj_yaw) P.traj_yaw[(size_t)q * (MAX_TRAJ + 1) + i] = st[ns];
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
      o.spec[0] = S.c_cand; o.spec[1] = S.c_cand - S.c_live; o.spec[2] = S.c_live; o.spec[3] = S.c_cut;
      o.t_begin = t_begin;
      o.t_end = wall_clock64();
      for (int i = 0; i < 10; i++) o.cyc[i] = S.cyc[i];
#ifdef MPLX_LOOKUP_TIMERS
      if (P.nq == 1 || q % 61 == 0) {  // (a batch: a sample of its queries -- a thousand workgroups printing at once tear each other's lines)
      printf("cyc2 q%d batches %llu:", q, S.cyc[7]);
      for (int i = 0; i < 24; i++) printf(" %llu", S.cyc2[i] / (S.cyc[7] ? S.cyc[7] : 1ull));
      printf("\n  max-over-lanes: heuristic done %llu, barrier arrival %llu\n", S.sum_heur / S.cyc[7], S.sum_arr / S.cyc[7]);
      printf("  ranking: mean near size %.1f, mean appended %.1f, batches on the all-pairs path %llu, batches with near > 256: %llu, of %llu\n", (double)S.dbg_n / S.cyc[7], (double)S.dbg_na / S.cyc[7], S.dbg_slow, S.dbg_n256, S.cyc[7]);
      for (int j = 0; j < 4; j++) { printf("  per-wave %d:", j); for (int w = 0; w < BLOCK / 64; w++) printf(" %llu", S.cycw[w][j] / (S.cyc[7] ? S.cyc[7] : 1ull)); printf("\n"); }
      }
#endif
    }
    // ---- pool recycling (SearchParams::chunk_bits): the finished query hands its chunks back to the pools
    bool recycled = false;
    if (P.chunk_bits) {
      recycled = true;
      if constexpr (HELP) {
        // Not while a helper may still be reading this query's records (it detaches when it sees the box inactive: the store ahead of
        // recoverTraj; in flight it has at most one wish list, ~100 us).  A helper that kept serving after the chunks had changed hands
        // could publish a row computed against THIS query's goal for a record that belongs to the next owner by then.  Bounded
        // (R4): a helper that does not leave keeps the chunks out of circulation for the rest of the launch, nothing else.
        if (P.boxes) {
          if (tid == 0) {
            int ok = 1;
            const HelpBox *box = P.boxes + blockIdx.x;
            for (uint32_t polls = 0; ld_u32(&box->helpers) != 0u; polls++) {
              if (polls > 40000u || ((polls & 1023u) == 1023u && guard_abort(P))) { ok = 0; break; }
              __builtin_amdgcn_s_sleep(32);
            }
            S.flag = ok;
          }
          __syncthreads();
          recycled = S.flag != 0;
          __syncthreads();
          if (recycled && searched) {  // the look-ahead records of its states: zero again for the next owner (written through: helpers on other XCDs read them)
            for (uint32_t ch = 0; ch < S.node_chunks; ch++) {
              double *base = (double *)(P.cache_c + ((size_t)Q.node_chunk(ch) << NODE_CH_LOG));
              static_assert(sizeof(CacheRec) == 16, "one 16-byte store per record");
              for (uint32_t i = tid; i < (1u << NODE_CH_LOG); i += BLOCK) st_f64x2_agent(base + 2 * (size_t)i, 0.0, 0.0);
            }
          }
        }
      }
      __syncthreads();  // (every wave's stores of this query -- and the zeroes above -- are acknowledged)
      // The chunks may go to a workgroup on ANOTHER XCD, and the XCDs' write-back L2s are not coherent with each other: what this
      // query left dirty in this XCD's L2 (its plain stores: records, predecessor entries, log entries) must reach memory NOW -- written
      // back later, by an eviction, it would land on top of the next owner's data (seen in the first version of this code: a recovered
      // trajectory that followed predecessor records of the chunk's previous life).  One agent-scope release (buffer_wbl2 sc1) per
      // finished query; the explicit wait is the guide's: the compiler may drop its own behind the write-back.
      if (recycled && tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      if (recycled) {
        for (uint32_t i = tid; i < S.node_chunks; i += BLOCK) Q.chunk_give(0, Q.node_chunk(i));
        for (uint32_t i = tid; i < S.edge_chunks; i += BLOCK) Q.chunk_give(1, Q.edge_chunk(i));
        for (uint32_t i = tid; i < S.open_chunks; i += BLOCK) Q.chunk_give(2, Q.open_chunk(i));
      }
    }
    // chunk tables of the query for the host's state-space getters (a recycling launch keeps no state space: the host refuses them)
    for (uint32_t i = tid; i < (uint32_t)MAX_NODE_CH; i += BLOCK)
      P.node_tables[(size_t)q * MAX_NODE_CH + i] = (i < S.node_chunks && !P.chunk_bits) ? Q.node_chunk(i) : NIL;
    for (uint32_t i = tid; i < (uint32_t)MAX_EDGE_CH; i += BLOCK)
      P.edge_tables[(size_t)q * MAX_EDGE_CH + i] = (i < S.edge_chunks && !P.chunk_bits) ? Q.edge_chunk(i) : NIL;
    if constexpr (HELP) {
      if (tid == 0) atomicAdd(P.done_word, 1ull);
    }
    __syncthreads();
  }
  if constexpr (HELP) {  // no query left to lead: help the leaders that are still running
    if (P.help_max > 0) {
      if (tid == 0) {
        S.help_idle = 0;
        S.help_quit = 0;
        // streamed batches: at most help_limit workgroups of the launch stay on as helpers; the others exit here, and the
        // workgroups of the next batch (another launch, another stream) take their compute units
        S.flag = (P.help_limit >= 0 && atomicAdd(P.cache_next + 4, 1u) >= (uint32_t)P.help_limit) ? 1 : 0;
      }
      __syncthreads();
      if (!S.flag) helper_loop<UL, K, CONTROL>(P, S, tid);
    }
  }
}

}  // namespace mplx
