// mplx_kernels.h -- gfx950 kernels of the motion-primitive search back-end.
//
//  * expand_kernel   : env_map::get_succ for K nodes, one workgroup per node (unit-test entry).
//  * astar_kernel    : GraphSearch::Astar resident on the device, one workgroup per in-flight query,
//                      queries pulled from a device counter (persistent workgroups).
//
// Work decomposition of one expansion (get_succ call sites: env_poly_map.h:45-69, env_cloud.h:50-70):
//   phase 1  lane = control input   : build primitive, end state, key, validate, sample count
//   phase 2  lane = (primitive, sample) pair, flattened over the workgroup: polynomial position ->
//            floatToInt -> one voxel byte; first blocked sample per primitive via LDS atomicMin
//   phase 3  lane = successor       : hash-dedup against the query's state space, relax, push
// OPEN is an exact min-priority structure under the strict total order (f, g, node id):
//   near set  = unsorted LDS array, popped by a workgroup-wide argmin
//   far set   = f-bucketed append-only log in HBM, NSUB linked sub-lists per bucket
// Stale entries (node since improved or closed) are dropped at pop time, which yields the same pop
// sequence as a decrease-key heap.
#pragma once
#include "mplx_device.h"

namespace mplx {

// Diagnostic / measurement switches of SearchParams::xflags (environment MPLX_X_FLAGS, read per launch): compiled out of the product
// (every one is a scalar test and a live SGPR on a query's serial chain); -DMPLX_DIAG_FLAGS=1 (tools/build_variant.sh diag) puts them
// back (round 5's tools/r05_jrk_batch.py / r05_ab.py, in the history).  Bit 8 -- the test-only spin the launch guard must end -- is always there.
//   1 table probes at agent scope   2 release / acquire fences around a look-ahead record   (4: round 5's doubled table, removed)
//   16 rows behind a release fence instead of the check word   32 TBL_DEAD_ID ahead of the parallel commit   64 claim wait in the
//   one-node kernels (on by default since round 5)   128 plain loads of sc1-stored state doubles   256 no look-ahead hit is taken
//   512 plain state stores   1024 rows unchecked   2048 no claim wait   4096 no TBL_DEAD_ID
#ifndef MPLX_DIAG_FLAGS
#define MPLX_DIAG_FLAGS 0
#endif
#define MPLX_XF(P, bit) (MPLX_DIAG_FLAGS && ((P).xflags & (bit)))
#ifndef MPLX_X_CLAIM_WAIT_1N
#define MPLX_X_CLAIM_WAIT_1N 1
#endif
// ------------------------------------------------------------------ small device helpers
__device__ __forceinline__ unsigned long long ld_u64(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_u64(unsigned long long *p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// state-table probe: a workgroup-scope load is a PLAIN load on gfx950 (non-tgsplit mode): it may hit in this compute
// unit's vector L1 and in the XCD's L2.  A slot of this query is only ever written by this workgroup, and it only goes
// EMPTY -> claim (compare-and-swap, executed at the memory side) -> entry (agent-scope store), never back; neither write
// updates the L1 copy of the line, so what a probe sees through the L1 may be one step behind:
//   * a stale EMPTY is caught by the claiming compare-and-swap, which returns what the slot really holds;
//   * a stale CLAIM of a slot that has its entry (the line was fetched between the two writes and survived in the L1)
//     would move the probe on and make the leader create the state a second time: the look-up re-reads a claim that
//     carries the query's tag with an agent-scope load before believing it (mplx_spec.h; round 3 -- it took the control
//     inputs moving to LDS, which stopped the churn of the L1, for this to show: the helper-assisted 125-input search then
//     created 3..120 duplicate states in half of its runs);
//   * a foreign entry, stale or not, just moves the probe on.
// The scheme needs the whole workgroup on one compute unit (one L1): see the guard below.
#if defined(__gfx950__) && defined(__AMDGCN_TGSPLIT__)
#error "libmplx assumes one workgroup = one compute unit (do not build with -mtgsplit): ld_u64_probe relies on it"
#endif
__device__ __forceinline__ unsigned long long ld_u64_probe(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t ld_u32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_u32(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// f64 through an agent-scope (sc1, write-through / L1-bypassing) 8-byte access: data another compute unit reads or wrote
__device__ __forceinline__ double ld_f64_agent(const double *p) { return __longlong_as_double((long long)ld_u64((const unsigned long long *)p)); }
__device__ __forceinline__ void st_f64_agent(double *p, double v) { st_u64((unsigned long long *)p, (unsigned long long)__double_as_longlong(v)); }
// two f64 through ONE 16-byte agent-scope (sc1, write-through) store: half the write-through transactions of two
// st_f64_agent (an 8-byte sc1 store is counted -- and carried -- as a partial-line write of its own).  p is 16-byte aligned.
// The `s_nop 1` INSIDE the string is not padding: a store of more than 8 bytes reads its data registers for two more cycles
// after it issues, and a vector instruction that overwrites them in that window changes what is stored.  For its own stores the
// compiler's hazard recogniser inserts the two wait states (gfx940+); it cannot see into an asm string, and the four registers
// of `v` are dead after the statement -- exactly what it reuses for the next pair.  Until round 5 the string ended with the
// store: under back-pressure from the memory pipeline (a neighbour's hipMemset, a saturated write queue) a node's state doubles
// reached memory wrong now and then, the node expanded into successors with other keys ("states created twice": more states,
// same edges), and a garbage position could ask for billions of samples -- the stall of round 4's driver run.  DESIGN.md 3.9.
__device__ __forceinline__ void st_f64x2_agent(double *p, double a, double b) {
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  const f64x2 v = {a, b};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
// ---- launch guard (mplx_device.h GuardBlock): the abort word and this workgroup's watch record live in host-coherent
// memory; system-scope accesses (sc0 sc1) go over the fabric every time.  A load costs a round trip to the host
// (microseconds): callers keep it off a query's serial chain or inside a loop that is waiting anyway.
constexpr int PLAN_ABORTED = 7;  // MPLX_PLAN_ABORTED
__device__ __forceinline__ bool guard_abort(const SearchParams &P) {
  return P.guard && __hip_atomic_load(&P.guard->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
}
__device__ __forceinline__ void guard_mark(const SearchParams &P, uint32_t phase, uint32_t q, unsigned long long count, unsigned long long info) {
  if (!P.guard) return;
  GuardRec *r = &P.guard->rec[blockIdx.x & (GUARD_SLOTS - 1)];
  __hip_atomic_store(&r->w1, info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&r->w0, ((unsigned long long)phase << 56) | ((unsigned long long)(q & 0xFFFFu) << 40) | (count & 0xFFFFFFFFFFull), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_SYSTEM);
}
// every GUARD_POLL_EVERY-th turn of a wait loop looks at the abort word (a wait loop sleeps ~1 us per turn)
constexpr uint32_t GUARD_POLL_EVERY = 1024;

__device__ __forceinline__ bool entry_less(double f1, double g1, uint32_t i1, double f2, double g2, uint32_t i2) {
  if (f1 != f2) return f1 < f2;
  if (g1 != g2) return g1 < g2;
  return i1 < i2;
}

// ---- cross-lane prefix operations on the VALU's DPP path (row shifts and row broadcasts: a few cycles per step)
// instead of ds_bpermute (__shfl_up: an LDS-crossbar round trip per step, ~30 dependent ones per batch of the
// search).  Every lane of the wave must be active.  WIDTH = 16, 32 or 64 consecutive lanes per segment.
// [row_shr:n = 0x110 + n, row_bcast:15 = 0x142 (into rows 1 and 3), row_bcast:31 = 0x143 (into rows 2 and 3): the
//  scan LLVM's own atomic optimiser emits for gfx9]
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t old, uint32_t src) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROW_MASK, 0xf, false);
}
template <int WIDTH>
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t x) {
  static_assert(WIDTH == 16 || WIDTH == 32 || WIDTH == 64, "segment of 16, 32 or 64 lanes");
  x += dpp_u32<0x111, 0xf>(0u, x);
  x += dpp_u32<0x112, 0xf>(0u, x);
  x += dpp_u32<0x114, 0xf>(0u, x);
  x += dpp_u32<0x118, 0xf>(0u, x);
  if constexpr (WIDTH >= 32) x += dpp_u32<0x142, 0xa>(0u, x);
  if constexpr (WIDTH >= 64) x += dpp_u32<0x143, 0xc>(0u, x);
  return x;
}
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long x) {  // source lane's value, 0 where there is none
  const uint32_t lo = dpp_u32<CTRL, 0xf>(0u, (uint32_t)x), hi = dpp_u32<CTRL, 0xf>(0u, (uint32_t)(x >> 32));
  return ((unsigned long long)hi << 32) | (unsigned long long)lo;
}
__device__ __forceinline__ unsigned long long row_incl_sum64(unsigned long long x) {  // rows of 16 lanes, mod 2^64
  x += dpp_u64<0x111>(x);
  x += dpp_u64<0x112>(x);
  x += dpp_u64<0x114>(x);
  x += dpp_u64<0x118>(x);
  return x;
}
__device__ __forceinline__ int row_incl_min(int x, int identity) {  // rows of 16 lanes
  int y;
  y = (int)dpp_u32<0x111, 0xf>((uint32_t)identity, (uint32_t)x); x = y < x ? y : x;
  y = (int)dpp_u32<0x112, 0xf>((uint32_t)identity, (uint32_t)x); x = y < x ? y : x;
  y = (int)dpp_u32<0x114, 0xf>((uint32_t)identity, (uint32_t)x); x = y < x ? y : x;
  y = (int)dpp_u32<0x118, 0xf>((uint32_t)identity, (uint32_t)x); x = y < x ? y : x;
  return x;
}
// minimum of x over the 64 lanes of the wave, in every lane (x >= 0 or +inf; every lane active): a DPP inclusive
// minimum scan (lanes without a source see +inf) and a read of lane 63
__device__ __forceinline__ double wave_min_f64(double x) {
  constexpr unsigned long long INF = 0x7FF0000000000000ull;
#define MPLX_MIN_STEP(CTRL, MASK)                                                                                        \
  {                                                                                                                      \
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);                                            \
    const uint32_t lo = dpp_u32<CTRL, MASK>((uint32_t)INF, (uint32_t)b), hi = dpp_u32<CTRL, MASK>((uint32_t)(INF >> 32), (uint32_t)(b >> 32)); \
    const double y = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));                              \
    x = y < x ? y : x;                                                                                                   \
  }
  MPLX_MIN_STEP(0x111, 0xf) MPLX_MIN_STEP(0x112, 0xf) MPLX_MIN_STEP(0x114, 0xf) MPLX_MIN_STEP(0x118, 0xf) MPLX_MIN_STEP(0x142, 0xa) MPLX_MIN_STEP(0x143, 0xc)
#undef MPLX_MIN_STEP
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)__double_as_longlong(x), 63);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)__double_as_longlong(x) >> 32), 63);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x) {
#define MPLX_MIN_STEP(CTRL, MASK) { const uint32_t y = dpp_u32<CTRL, MASK>(0xFFFFFFFFu, x); x = y < x ? y : x; }
  MPLX_MIN_STEP(0x111, 0xf) MPLX_MIN_STEP(0x112, 0xf) MPLX_MIN_STEP(0x114, 0xf) MPLX_MIN_STEP(0x118, 0xf) MPLX_MIN_STEP(0x142, 0xa) MPLX_MIN_STEP(0x143, 0xc)
#undef MPLX_MIN_STEP
  return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
// The wave's smallest OPEN entry under entry_less -- (f, g, id) in lexicographic order; `have` false: the lane holds
// none -- delivered to every lane; false when no lane holds one.  One DPP minimum per level, further levels only among
// the lanes that tie (instead of six butterfly steps of ds_bpermute on all four fields).
__device__ __forceinline__ bool wave_min_entry(bool have, double &f, double &g, uint32_t &id, uint32_t &pos) {
  unsigned long long m = __ballot(have);
  if (!m) return false;
  if (m & (m - 1ull)) {
    const double fm = wave_min_f64(have ? f : INFINITY);
    m = __ballot(have && f == fm);
    if (m & (m - 1ull)) {
      const bool tie = have && f == fm;
      const double gm = wave_min_f64(tie ? g : INFINITY);
      m = __ballot(tie && g == gm);
      if (m & (m - 1ull)) {
        const bool tie2 = tie && g == gm;
        const uint32_t im = wave_min_u32(tie2 ? id : 0xFFFFFFFFu);
        m = __ballot(tie2 && id == im);
      }
    }
  }
  const int w = __ffsll((long long)m) - 1;
  const unsigned long long fb = (unsigned long long)__double_as_longlong(f), gb = (unsigned long long)__double_as_longlong(g);
  const uint32_t f0 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)fb, w), f1 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(fb >> 32), w);
  const uint32_t g0 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)gb, w), g1 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(gb >> 32), w);
  f = __longlong_as_double((long long)(((unsigned long long)f1 << 32) | f0));
  g = __longlong_as_double((long long)(((unsigned long long)g1 << 32) | g0));
  id = (uint32_t)__builtin_amdgcn_readlane((int)id, w);
  pos = (uint32_t)__builtin_amdgcn_readlane((int)pos, w);
  return true;
}
// value held by the last lane of my segment (uniform per segment)
template <int WIDTH>
__device__ __forceinline__ uint32_t segment_last(uint32_t x, int lane) {
  if constexpr (WIDTH == 64) {
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
  } else {
    static_assert(WIDTH == 32, "two segments per wave");
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)x, 31), b = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
    return (lane & 32) ? b : a;
  }
}

struct alignas(8) OpenRec {  // OPEN-log record (OPEN_BYTES)
  double f, g;
  uint32_t id, next;
};
struct EdgeRec {  // predecessor record (EDGE_BYTES)
  uint32_t parent, next, action;
};
static_assert(sizeof(OpenRec) == OPEN_BYTES && sizeof(EdgeRec) == EDGE_BYTES, "record sizes");

// The query's three chunk tables live in LDS whole (33 M states, 134 M predecessor records, 33 M log entries per query: 8 KB).
template <int BLOCK, int KUNITS = 1, int NCAP_ = NC>
struct Smem {
  static constexpr int NCAP = NCAP_;  // near-set capacity
  // OPEN near set
  double near_f[NCAP_], near_g[NCAP_];
  uint32_t near_id[NCAP_], near_idx[NCAP_];
  uint32_t cnt[2][NB];  // entries per bucket: [0] fine level (inside coarse bucket cur1), [1] coarse level
  // chunk tables of the running query
  uint16_t node_tbl[MAX_NODE_CH], edge_tbl[MAX_EDGE_CH], open_tbl[MAX_OPEN_CH];  // pool chunk ids (the host keeps pools below 65536 chunks)
  // expansion scratch
  // pre-divided non-zero polynomial coefficients (pack_q_c) of a sample's primitive, split into the
  // part that only depends on the control input (per query) and the part that only depends on the
  // node being expanded (per unit): no per-primitive staging
  double uq[3][BLOCK / KUNITS];  // q[0] of control input i per axis: U[i][ax] / {1, 2, 6, 24}
  // the control inputs by axis and their edge costs (fill_uq): P.U / P.ucost are global loads at the head of every
  // expansion / before every commit otherwise -- an L2 round trip on a query's serial chain (lane_u)
  double u_lds[3][BLOCK / KUNITS], ucost_lds[BLOCK / KUNITS];
  double qn[KUNITS][3][5];       // q[1..] of the node per axis
  double dts[BLOCK];
  // per unit: flattened sample e -> (primitive << 8) | sample index; 12 samples per lane of the unit
  // (at least 576 = 27 x 21 + 9) before the generic loop takes over
  uint16_t owner[KUNITS][KUNITS > 1 ? (BLOCK / KUNITS * 12 > 576 ? BLOCK / KUNITS * 12 : 576) : OWN];
  int32_t slow[KUNITS];  // unit has more samples than the owner map covers -> generic sample loop
  uint32_t node_blk[KUNITS];  // the unit's node cell (sample 0 of every primitive): 0 free, 1 occupied, 2 outside the map
  uint32_t offs[KUNITS][BLOCK / KUNITS + 1];
  // look-ahead cache entry of the unit's node, if a helper workgroup left one (c_row 0: none)
  uint32_t hc_row[KUNITS], hc_valid[KUNITS], hc_blocked[KUNITS], hc_reads[KUNITS];
  uint32_t blk[BLOCK];  // first blocked sample: (i << 1) | inside
  unsigned long long dupset[KUNITS > 1 ? 2 : 2 * BLOCK];  // unused by the multi-unit kernel
  double cur[KUNITS][13];       // state of the node(s) being expanded (p,v,a,j,t)
  int32_t cur_key[KUNITS][MAX_KEY];
  HeurParams hp;
  // scan / reduce scratch
  uint32_t wsum[BLOCK / 64 + 1];
  double red_f[BLOCK / 64], red_g[BLOCK / 64];
  uint32_t red_id[BLOCK / 64], red_pos[BLOCK / 64];
  uint32_t hist[64];
  // scalars
  uint32_t n_near, n_nodes, n_edges, n_log;
  uint32_t reserve;  // near-set slots one iteration may need (successors pushed + candidates returned)
  uint32_t node_chunks, edge_chunks, open_chunks;  // chunks owned
  int32_t cur1, cur0;  // active coarse bucket, active fine bucket inside it
  int32_t pull_n, pull_list[MERGE_MAXB_ALL];  // the run of fine buckets one refill pulls together (refill / pull_fine_run)
  double lo1;          // f of the lower edge of coarse bucket cur1
  double ts_f, ts_g;
  uint32_t ts_id;      // split threshold inside fine bucket cur0
  double f_base;
  uint32_t cur_id;
  double cur_g;
  int32_t status, flag, q_index;
  uint32_t tmp_u;
  double tmp_d0;
  unsigned long long c_expanded, c_closed, c_prims, c_succ, c_succ_finite, c_reads, c_push, c_reopen, c_refill, c_evict, c_hash;
  unsigned long long c_cand, c_live, c_cut;  // (speculative kernels) candidates taken, live units expanded, expanded units cut and returned to OPEN
  unsigned long long cyc[10];
  double cur_yaw[KUNITS];  // yaw of the node(s) being expanded (yaw-carrying searches)
  __device__ __forceinline__ uint32_t *pot_scratch() { return (uint32_t *)dupset; }
};

// Per-phase cycle counters of the search kernels (mplx_result_cycles): s_memtime + a 64-bit LDS add on thread 0 at every
// phase boundary sit on the serial chain of a query (about eight per batch, ~1 k cycles of 41 k), so the product build
// leaves them out and reports zeros for the time fields (the event counts -- batches, look-ahead hits -- stay);
// -DMPLX_PHASE_TIMERS=1 (tools/build_variant.sh timers) puts them back.
#ifndef MPLX_PHASE_TIMERS
#define MPLX_PHASE_TIMERS 0
#endif
#if MPLX_PHASE_TIMERS || defined(MPLX_LOOKUP_TIMERS) || defined(MPLX_FINE_TIMERS)
#define MPLX_TIC(var) const unsigned long long var = __builtin_readcyclecounter()
#define MPLX_TOC(S, k, var) do { if (threadIdx.x == 0) (S).cyc[k] += __builtin_readcyclecounter() - (var); } while (0)
#else
#define MPLX_TIC(var) [[maybe_unused]] constexpr unsigned long long var = 0ull
#define MPLX_TOC(S, k, var) do { } while (0)
#endif

// The value, made opaque to the optimiser (no instruction).  The search kernels sit at their register limit, and the
// compiler hoists the LDS addresses of small per-lane arrays (&S.x[tid]: one add) out of the batch loop, then spills them:
// every use became a scratch load plus s_waitcnt vmcnt(0) -- a wait for everything the wave has in flight -- on the serial
// chain of a query (five in a row in the batch set-up, four in the cut evaluation: ~2 k cycles each section).  An index
// that went through here is recomputed where it is used instead.
__device__ __forceinline__ int opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}

template <int BLOCK, class SM>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, SM &S, int tid, uint32_t &total) {
  const int lane = tid & 63, wave = tid >> 6;
  const uint32_t x = wave_incl_sum<64>(v);
  if constexpr (BLOCK == 64) {
    total = segment_last<64>(x, lane);
    return x - v;
  } else {
    if (lane == 63) S.wsum[wave] = x;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; w++) {
      uint32_t s = S.wsum[w];
      if (w < wave) base += s;
      tot += s;
    }
    total = tot;
    __syncthreads();
    return base + x - v;
  }
}

template <int BLOCK, class SM>
__device__ __forceinline__ bool block_any(bool p, SM &S, int tid) {
  unsigned long long b = __ballot(p);
  if constexpr (BLOCK == 64) {
    return b != 0ull;
  } else {
    if ((tid & 63) == 0) S.wsum[tid >> 6] = b != 0ull;
    __syncthreads();
    bool r = false;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; w++) r = r || (S.wsum[w] != 0);
    __syncthreads();
    return r;
  }
}

// Synchronise the lanes of one expansion unit.  A unit of <= 64 lanes lives inside one wave, whose
// LDS operations execute in program order: ordering them is enough and no workgroup barrier is
// needed, so units never wait for each other inside the expansion.  Larger units span waves.
template <int UL>
__device__ __forceinline__ void unit_sync() {
  if constexpr (UL <= 64) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}

// exclusive scan inside an expansion unit of UL lanes (a unit is UL/64 consecutive waves); every
// thread of the workgroup must call it (one workgroup barrier when a unit spans several waves)
template <int UL, int BLOCK, class SM>
__device__ __forceinline__ uint32_t unit_excl_scan(uint32_t v, SM &S, int tid, uint32_t &total) {
  if constexpr (UL == BLOCK) {
    return block_excl_scan<BLOCK>(v, S, tid, total);
  } else {
    if constexpr (UL < 64) {  // several units per wave: segmented scan, no barrier
      const uint32_t x = wave_incl_sum<UL>(v);
      total = segment_last<UL>(x, tid & 63);
      return x - v;
    }
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t x = wave_incl_sum<64>(v);
    if constexpr (UL == 64) {
      total = segment_last<64>(x, lane);
      return x - v;
    } else {
      if (lane == 63) S.wsum[wave] = x;
      __syncthreads();
      constexpr int WPU = UL / 64 > 0 ? UL / 64 : 1;
      const int w0 = (wave / WPU) * WPU;
      uint32_t base = 0, tot = 0;
#pragma unroll
      for (int w = 0; w < WPU; w++) {
        uint32_t t = S.wsum[w0 + w];
        if (w0 + w < wave) base += t;
        tot += t;
      }
      total = tot;
      __syncthreads();
      return base + x - v;
    }
  }
}

__device__ __forceinline__ int bucket_idx(double f, double lo, double width) {
  double b = floor((f - lo) / width);
  if (!(b > 0.0)) return 0;  // also NaN
  if (b >= (double)(NB - 1)) return NB - 1;
  return (int)b;
}

// ------------------------------------------------------------------ expansion: phases 1 and 2
struct LaneSucc {
  State tn;
  int32_t key[MAX_KEY];
  bool valid;    // successor emitted
  bool blocked;  // is_free(pr) failed -> cost inf
  uint32_t reads;
  uint32_t pot;  // (POT) sum of the potential over the primitive's samples
  double yaw;    // (YAW) successor yaw, normalised, and its key integer round(yaw / 0.1)
  int32_t yaw_key;
};

// One expansion unit = UL consecutive threads expanding one node (unit index ku, lane index lu inside
// the unit); the workgroup holds BLOCK / UL units.  `live` false: the unit idles (it still takes part
// in the workgroup barriers).
struct NoHook {
  __device__ __forceinline__ void operator()(const LaneSucc &) const {}
};
// control input `lu`, axis `ax`: from LDS in the kernels that stage the lattice there (SmemSpec::u_lds: P.U is a global
// load at the head of every expansion otherwise -- an L2 round trip on a query's serial chain), from P.U elsewhere
template <class SM>
__device__ __forceinline__ auto lane_u(const SM &S, const SearchParams &, int lu, int ax, int) -> decltype((void)S.u_lds, double()) { return S.u_lds[ax][lu]; }
template <class SM>
__device__ __forceinline__ double lane_u(const SM &, const SearchParams &P, int lu, int ax, long) { return P.U[3 * lu + ax]; }
// `after_phase1(L)` runs once the successor state and key of the lane's primitive are known (L.valid),
// before the voxel sampling: the caller can start memory traffic that depends on the key only.
// CACHE: units whose S.hc_row is non-zero take validity / blocked flags from the look-ahead cache entry
// (S.hc_valid, S.hc_blocked) and skip validate_primitive and the voxel sampling altogether.
// POT: the auxiliary map (potential field / search region, MapDev::aux) is read next to the occupancy when it exists:
// a sample outside the search region blocks the primitive, the potential of every sample is summed into L.pot.  Only
// the one-unit kernels (UL == BLOCK) pass it; the speculative kernels are compiled without.
// YAW: the states carry yaw (use_yaw lattices): successor yaw = normalise(yaw + u_yaw dt), its key integer takes part in
// "tn == curr", and validate_yaw joins validate_primitive (mplx_math.h).  One-unit kernels only.
template <int UL, int BLOCK, int CONTROL, bool CACHE = false, bool POT = false, bool YAW = false, class SM, class Hook = NoHook>
__device__ __forceinline__ void expand_unit(const SearchParams &P, SM &S, int tid, bool live_unit, LaneSucc &L, Hook after_phase1 = Hook()) {
  constexpr int NQ = nq_c(CONTROL);
  const int8_t *__restrict__ aux = nullptr;
  uint32_t *pots = nullptr;  // per-primitive potential sums, one word per lane, in LDS that is idle during the expansion
  if constexpr (POT) {       // (the one-unit kernels: the commit's duplicate-key set; the speculative kernels: a batch-table column)
    aux = P.map.aux;
    pots = S.pot_scratch();
    if (aux) pots[tid] = 0;
  }
  const int ku = tid / UL, lu = tid % UL;
  const double T = P.dt;
  bool cached = false;
  if constexpr (CACHE) cached = live_unit && S.hc_row[ku] != 0u;
#ifdef MPLX_FINE_TIMERS
  MPLX_TIC(tf0);
#endif
  L.valid = false;
  L.blocked = false;
  L.reads = 0;
  L.pot = 0;
  uint32_t my_cnt = 0, my_pairs = 0;
  if (live_unit && lu < P.n_u) {
    double c[3][6];
#pragma unroll
    for (int ax = 0; ax < 3; ax++) prim_build_axis(CONTROL, S.cur[ku][ax], S.cur[ku][3 + ax], S.cur[ku][6 + ax], S.cur[ku][9 + ax], lane_u(S, P, lu, ax, 0), c[ax]);
#pragma unroll
    for (int ax = 0; ax < 3; ax++) {
      L.tn.p[ax] = pos_at_c<CONTROL>(c[ax], T);
      L.tn.v[ax] = vel_at_c<CONTROL>(c[ax], T);
      L.tn.a[ax] = acc_at_c<CONTROL>(c[ax], T);
      L.tn.j[ax] = jrk_at_c<CONTROL>(c[ax], T);
    }
    state_key_c<CONTROL>(L.tn, L.key);
    // (key comparisons are written without short circuit everywhere: `a && b` makes the compiler wait for each word of
    // the other key -- an LDS or memory round trip -- before it asks for the next)
    uint32_t kdiff = 0;
#pragma unroll
    for (int i = 0; i < key_len_c(CONTROL); i++) kdiff |= (uint32_t)(L.key[i] ^ S.cur_key[ku][i]);
    bool same = kdiff == 0u;
    bool yaw_ok = true;
    if constexpr (YAW) {  // (one-unit kernels, and the YAW builds of the speculative kernel: S.cur_yaw is per unit)
      const double yaw0 = S.cur_yaw[ku], uy = P.U_yaw ? P.U_yaw[lu] : 0.0;
      L.yaw = normalize_yaw((uy * T + 0.0) + yaw0);  // the VEL-type yaw channel evaluated at T, normalised like evaluate()
      L.yaw_key = (int32_t)round(L.yaw / KEY_RES_YAW);
      same = same && L.yaw_key == (int32_t)round(yaw0 / KEY_RES_YAW);
      if (P.yaw_max > 0)  // validate_yaw: both ends, planar velocity against the yaw direction
        yaw_ok = yaw_end_ok(vel_at_c<CONTROL>(c[0], 0.0), vel_at_c<CONTROL>(c[1], 0.0), normalize_yaw(yaw0 + 0.0), P.yaw_cos) && yaw_end_ok(L.tn.v[0], L.tn.v[1], L.yaw, P.yaw_cos);
    }
    double max_v = 0.0;
    bool ok;
    if (cached) {  // decided ahead of time by a helper workgroup: nothing left to sample
      if constexpr (UL <= 64) {
        L.valid = (S.hc_valid[ku] >> lu) & 1u;
        L.blocked = (S.hc_blocked[ku] >> lu) & 1u;
      } else {  // large lattice: the mask words are at the head of the helper's row
        const uint32_t *rw = (const uint32_t *)(P.cache_h + (size_t)(S.hc_row[ku] - 1u) * cache_row_doubles(UL));
        L.valid = (ld_u32(rw + (lu >> 5)) >> (lu & 31)) & 1u;
        L.blocked = (ld_u32(rw + 4 + (lu >> 5)) >> (lu & 31)) & 1u;
      }
      ok = false;
    } else {
#ifdef MPLX_GENERIC_VALIDATE
      ok = !same && validate_and_maxv(CONTROL, c, T, P.v_max, P.a_max, P.j_max, &max_v);
#else
      ok = !same && validate_and_maxv_c<CONTROL>(c, T, P.v_max, P.a_max, P.j_max, &max_v);
#endif
      ok = ok && yaw_ok;
    }
    if (ok) {
      int n = (int)ceil(max_v * T / P.map.res);
      my_cnt = (uint32_t)(n + 1);
      my_pairs = (uint32_t)n;  // sample 0 is the node itself: tested once per unit, below
      S.dts[tid] = n > 0 ? T / n : 0.0;
      L.valid = true;
    }
  }
  after_phase1(L);
  if (live_unit && lu < 3) {  // node part of the pre-divided coefficients of axis lu
    double c0[6], qc[5];
    prim_build_axis(CONTROL, S.cur[ku][lu], S.cur[ku][3 + lu], S.cur[ku][6 + lu], S.cur[ku][9 + lu], 0.0, c0);
    pack_q_c<CONTROL>(c0, qc);
#pragma unroll
    for (int k = 1; k < NQ; k++) S.qn[ku][lu][k] = qc[k];
  }
#ifdef MPLX_FINE_TIMERS
  MPLX_TOC(S, 3, tf0);
  MPLX_TIC(tf1);
#endif
  // Sample 0 of every primitive is the node position (p(0) = c5): one cell test per unit instead of
  // n_u, by a lane that has no primitive when there is one.  The load is consumed after phase 2.
  const int nl = P.n_u < UL ? P.n_u : 0;
  uint32_t node_code = 0;
  if (live_unit && lu == nl && !cached) {
    int32_t c[3];
    bool in = true;
#pragma unroll
    for (int ax = 0; ax < 3; ax++) {
      c[ax] = float_to_cell(S.cur[ku][ax], P.map.origin[ax], P.map.res);
      in = in && c[ax] >= 0 && c[ax] < P.map.dim[ax];
    }
    if (in) {
      const uint32_t brick = (uint32_t)(c[0] >> 3) + (uint32_t)P.map.nb[0] * ((uint32_t)(c[1] >> 3) + (uint32_t)P.map.nb[1] * (uint32_t)(c[2] >> 3));
      const uint32_t bit = (uint32_t)(c[0] & 7) | ((uint32_t)(c[1] & 7) << 3) | ((uint32_t)(c[2] & 7) << 6);
      node_code = (P.map.bricks[brick * 16u + (bit >> 5)] >> (bit & 31u)) & 1u;
      if constexpr (POT) {
        if (aux && !node_code) {
          const int av = aux[(size_t)c[0] + (size_t)P.map.dim[0] * c[1] + (size_t)P.map.dim[0] * P.map.dim[1] * c[2]];
          if (av < 0) node_code = 1;                     // the node's own cell is outside the search region (after one read)
          else node_code = (uint32_t)av << 8;            // its potential counts for every primitive (sample 0)
        }
      }
    } else {
      node_code = 2;
    }
  }
  S.blk[tid] = 0xFFFFFFFFu;
  uint32_t total;
  uint32_t off = unit_excl_scan<UL, BLOCK>(my_pairs, S, tid, total);
  S.offs[ku][lu] = off;
  if (lu == UL - 1) S.offs[ku][UL] = total;
  // owner map: flattened sample e -> (primitive, sample index); one LDS read replaces a search
  constexpr uint32_t OWNU = sizeof(S.owner[0]) / sizeof(uint16_t);
  if (lu == 0) S.slow[ku] = 0;
  for (uint32_t i = 0; i < my_pairs && i < 255u && off + i < OWNU; i++) S.owner[ku][off + i] = (uint16_t)((lu << 8) | (i + 1));
  unit_sync<UL>();
  if (my_pairs > 255u || (lu == UL - 1 && total > OWNU)) S.slow[ku] = 1;
  unit_sync<UL>();
#ifdef MPLX_FINE_TIMERS
  MPLX_TOC(S, 5, tf1);
  MPLX_TIC(tf2);
#endif
  // phase 2: flattened (primitive, sample) pairs, UNR voxel loads in flight per lane
  const int8_t *__restrict__ map = P.map.data;
  const uint32_t *__restrict__ bricks = P.map.bricks;
  const int dx = P.map.dim[0], dy = P.map.dim[1], dz = P.map.dim[2];
  const int nb0 = P.map.nb[0], nb1 = P.map.nb[1];
  constexpr int UNR = 4;  // (round 6: 6 / 8 in flight measured 3 % / 10 % slower -- spilled registers)
  if (!S.slow[ku]) {
    // Staged over the UNR pairs of a lane (all LDS reads of a stage are independent, one wait per
    // stage), branch-free: dead slots recompute the last live pair, outside cells read voxel 0, and
    // both are masked when the result is recorded.
    double qnr[3][NQ];  // node part of the coefficients: loop invariant
#pragma unroll
    for (int ax = 0; ax < 3; ax++)
#pragma unroll
      for (int k = 1; k < NQ; k++) qnr[ax][k] = S.qn[ku][ax][k];
    const double ox = P.map.origin[0], oy = P.map.origin[1], oz = P.map.origin[2], rs = P.map.res;
    const double irs = 1.0 / rs;
    for (uint32_t e0 = lu; e0 < total; e0 += UL * UNR) {
      uint32_t o[UNR];
#pragma unroll
      for (int r = 0; r < UNR; r++) {
        const uint32_t e = e0 + r * UL;
        o[r] = S.owner[ku][e < total ? e : total - 1];
      }
      double dtr[UNR], u0[UNR], u1[UNR], u2[UNR];
#pragma unroll
      for (int r = 0; r < UNR; r++) {
        const int pl = (int)(o[r] >> 8);
        dtr[r] = S.dts[ku * UL + pl];
        u0[r] = S.uq[0][pl];
        u1[r] = S.uq[1][pl];
        u2[r] = S.uq[2][pl];
      }
      int32_t cx[UNR], cy[UNR], cz[UNR];
#pragma unroll
      for (int r = 0; r < UNR; r++) {
        const double t = (double)(o[r] & 255u) * dtr[r];
        double qq[NQ];
#pragma unroll
        for (int k = 1; k < NQ; k++) qq[k] = qnr[0][k];
        qq[0] = u0[r];
        cx[r] = float_to_cell_inv(pos_at_qc_cell<CONTROL>(qq, t), ox, rs, irs);
#pragma unroll
        for (int k = 1; k < NQ; k++) qq[k] = qnr[1][k];
        qq[0] = u1[r];
        cy[r] = float_to_cell_inv(pos_at_qc_cell<CONTROL>(qq, t), oy, rs, irs);
#pragma unroll
        for (int k = 1; k < NQ; k++) qq[k] = qnr[2][k];
        qq[0] = u2[r];
        cz[r] = float_to_cell_inv(pos_at_qc_cell<CONTROL>(qq, t), oz, rs, irs);
      }
      int32_t vv[UNR];
      bool inside[UNR];
#pragma unroll
      for (int r = 0; r < UNR; r++) {
        inside[r] = !(cx[r] < 0 || cx[r] >= dx || cy[r] < 0 || cy[r] >= dy || cz[r] < 0 || cz[r] >= dz);
        const int sx = min(max(cx[r], 0), dx - 1), sy = min(max(cy[r], 0), dy - 1), sz = min(max(cz[r], 0), dz - 1);
        // occupancy bit from the bricked bitmap (always a valid address)
        // (32-bit word index: the host refuses maps with more than 2^26 bricks)
        const uint32_t brick = (uint32_t)(sx >> 3) + (uint32_t)nb0 * ((uint32_t)(sy >> 3) + (uint32_t)nb1 * (uint32_t)(sz >> 3));
        const uint32_t bit = (uint32_t)(sx & 7) | ((uint32_t)(sy & 7) << 3) | ((uint32_t)(sz & 7) << 6);
        vv[r] = (int32_t)((bricks[brick * 16u + (bit >> 5)] >> (bit & 31u)) & 1u);
        if constexpr (POT) {
          if (aux) {  // (clamped address; masked below when the sample is outside)
            const int av = aux[(size_t)sx + (size_t)dx * sy + (size_t)dx * dy * sz];
            if (av < 0) vv[r] = 1;                                  // outside the search region: blocked, like an occupied voxel
            else if (!vv[r]) vv[r] = -av;                           // free: carry the potential (as a non-positive number)
          }
        }
      }
#pragma unroll
      for (int r = 0; r < UNR; r++) {
        if (e0 + r * UL < total) {
          const int pc = ku * UL + (int)(o[r] >> 8);
          const uint32_t i = o[r] & 255u;
          if (!inside[r])
            atomicMin(&S.blk[pc], i << 1);
          else if (vv[r] > 0)
            atomicMin(&S.blk[pc], (i << 1) | 1u);
          else if (POT && vv[r] < 0)
            atomicAdd(&pots[pc], (uint32_t)(-vv[r]));
        }
      }
    }
  } else {
    for (uint32_t e = lu; e < total; e += UL) {  // generic: search the primitive, any sample count
      int lo = 0, hi = UL;  // largest p with offs[p] <= e
      while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (S.offs[ku][mid] <= e) lo = mid; else hi = mid;
      }
      const int pl = lo, pc = ku * UL + pl;
      const uint32_t i = e - S.offs[ku][pl] + 1u;  // samples 1..n (sample 0: the node cell test)
      const double t = (double)i * S.dts[pc];
      int32_t cell[3];
#pragma unroll
      for (int ax = 0; ax < 3; ax++) {
        double qq[NQ];
        qq[0] = S.uq[ax][pl];
#pragma unroll
        for (int k = 1; k < NQ; k++) qq[k] = S.qn[ku][ax][k];
        cell[ax] = float_to_cell(pos_at_qc<CONTROL>(qq, t), P.map.origin[ax], P.map.res);
      }
      if (cell[0] < 0 || cell[0] >= dx || cell[1] < 0 || cell[1] >= dy || cell[2] < 0 || cell[2] >= dz)
        atomicMin(&S.blk[pc], i << 1);
      else if (map[(size_t)cell[0] + (size_t)dx * cell[1] + (size_t)dx * dy * cell[2]] > 0)
        atomicMin(&S.blk[pc], (i << 1) | 1u);
      else if constexpr (POT) {
        if (aux) {
          const int av = aux[(size_t)cell[0] + (size_t)dx * cell[1] + (size_t)dx * dy * cell[2]];
          if (av < 0) atomicMin(&S.blk[pc], (i << 1) | 1u);
          else if (av > 0) atomicAdd(&pots[pc], (uint32_t)av);
        }
      }
    }
  }
#ifdef MPLX_FINE_TIMERS
  MPLX_TOC(S, 9, tf2);   // thread 0's own phase-2 work, before waiting for the other units
#endif
  if (live_unit && lu == nl) S.node_blk[ku] = node_code;
  unit_sync<UL>();
  if (L.valid && !cached) {
    uint32_t code = S.blk[tid];
    const uint32_t nb = S.node_blk[ku] & 0xFFu;
    if (nb) code = nb == 2u ? 0u : 1u;  // blocked at sample 0 (outside: no voxel read)
    L.blocked = code != 0xFFFFFFFFu;
    L.reads = L.blocked ? (code >> 1) + (code & 1u) : my_cnt;
    if constexpr (POT) {
      if (aux && !L.blocked) L.pot = pots[tid] + (S.node_blk[ku] >> 8);
    }
  }
}

// control-input part of the pre-divided coefficients (per launch configuration)
template <int BLOCK, int CONTROL, class SM>
__device__ __forceinline__ void fill_uq(const SearchParams &P, SM &S, int tid) {
  for (int i = tid; i < 3 * P.n_u; i += BLOCK) {
    const int pi = i / 3, ax = i % 3;
    double c0[6], qc[5];
    prim_build_axis(CONTROL, 0.0, 0.0, 0.0, 0.0, P.U[3 * pi + ax], c0);
    pack_q_c<CONTROL>(c0, qc);
    S.uq[ax][pi] = qc[0];
    S.u_lds[ax][pi] = P.U[3 * pi + ax];
  }
  for (int i = tid; i < P.n_u; i += BLOCK) S.ucost_lds[i] = P.ucost[i];
}

// ------------------------------------------------------------------ expand_kernel (unit-test entry)
// node_yaw (YAW): yaw of each node
template <int BLOCK, int CONTROL, bool YAW = false>
__global__ __launch_bounds__(BLOCK) void expand_kernel(SearchParams P, const State *nodes, const double *node_t, int K, SuccOut *out, const double *node_yaw = nullptr) {
  __shared__ Smem<BLOCK> S;
  const int tid = threadIdx.x;
  fill_uq<BLOCK, CONTROL>(P, S, tid);
  for (int k = blockIdx.x; k < K; k += gridDim.x) {
    if (tid < 12) S.cur[0][tid] = ((const double *)&nodes[k])[tid];
    if (tid == 12) S.cur[0][12] = node_t[k];
    if (YAW && tid == 13) S.cur_yaw[0] = node_yaw[k];
    __syncthreads();
    if (tid == 0) {
      State s;
      for (int i = 0; i < 12; i++) ((double *)&s)[i] = S.cur[0][i];
      state_key_c<CONTROL>(s, S.cur_key[0]);
    }
    __syncthreads();
    LaneSucc L;
    expand_unit<BLOCK, BLOCK, CONTROL, false, true, YAW>(P, S, tid, true, L);
    if (tid < P.n_u) {
      SuccOut &o = out[(size_t)k * P.n_u + tid];
      for (int ax = 0; ax < 3; ax++) {
        o.pos[ax] = L.tn.p[ax];
        o.vel[ax] = L.tn.v[ax];
        o.acc[ax] = L.tn.a[ax];
        o.jrk[ax] = L.tn.j[ax];
      }
      o.yaw = YAW ? L.yaw : 0.0;
      o.t = S.cur[0][12] + P.dt;
      o.control = YAW ? (P.control | CTRL_YAW_BIT) : P.control;
      o.enable_t = 0;
      o.cost = L.valid ? (L.blocked ? INFINITY : (P.map.aux ? P.ucost[tid] + P.pot_weight * (double)L.pot : P.ucost[tid])) : 0.0;
      o.action = tid;
      o.valid = L.valid ? 1 : 0;
#pragma unroll
      for (int i = 0; i < 12; i++) o.key[i] = i < key_len_c(CONTROL) ? L.key[i] : 0;
      o.nkey = P.nk;
      if constexpr (YAW) {
        if (key_len_c(CONTROL) < 12) o.key[key_len_c(CONTROL)] = L.yaw_key;  // (SNP + yaw: 13 integers, the yaw key is not reported here)
        o.nkey = P.nk + 1;
      }
      o.voxel_reads = (int32_t)L.reads;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ chunked pool access
template <int BLOCK, int CONTROL, class SM = Smem<BLOCK>>
struct QView {
  const SearchParams &P;
  SM &S;
  uint32_t *bkt_head;
  // pool chunk of the query's c-th node / predecessor / OPEN-log chunk
  __device__ __forceinline__ uint32_t node_chunk(uint32_t c) const { return S.node_tbl[c]; }
  __device__ __forceinline__ uint32_t edge_chunk(uint32_t c) const { return S.edge_tbl[c]; }
  __device__ __forceinline__ uint32_t open_chunk(uint32_t c) const { return S.open_tbl[c]; }
  __device__ __forceinline__ char *node(uint32_t i) const {
    return P.node_pool + (((size_t)node_chunk(i >> NODE_CH_LOG) << NODE_CH_LOG) + (i & ((1u << NODE_CH_LOG) - 1))) * rec_bytes(CONTROL);
  }
  // index of node i's record in the shared pool (the look-ahead cache is indexed by it)
  __device__ __forceinline__ uint32_t node_rec(uint32_t i) const {
    return (node_chunk(i >> NODE_CH_LOG) << NODE_CH_LOG) + (i & ((1u << NODE_CH_LOG) - 1));
  }
  __device__ __forceinline__ EdgeRec *edge(uint32_t i) const {
    return (EdgeRec *)(P.edge_pool + (((size_t)edge_chunk(i >> EDGE_CH_LOG) << EDGE_CH_LOG) + (i & ((1u << EDGE_CH_LOG) - 1))) * EDGE_BYTES);
  }
  __device__ __forceinline__ OpenRec *open(uint32_t i) const {
    return (OpenRec *)(P.open_pool + (((size_t)open_chunk(i >> OPEN_CH_LOG) << OPEN_CH_LOG) + (i & ((1u << OPEN_CH_LOG) - 1))) * OPEN_BYTES);
  }
  // (one thread) a free chunk of pool `pool` from the recycling bitmap (set bit: free), NIL when there is none.  The scan starts at a
  // word that differs from workgroup to workgroup; a chunk is taken with atomicAnd -- whoever sees the bit in the returned word owns it.
  // Rare (once per 32 768 states): no attempt at speed.
  __device__ __forceinline__ uint32_t chunk_take(int pool) const {
    uint32_t *bits = P.chunk_bits + P.chunk_word0[pool];
    const uint32_t nw = P.chunk_words[pool];
    const uint32_t w0 = (blockIdx.x * 2654435761u) % nw;
    for (uint32_t k = 0; k < nw; k++) {
      const uint32_t w = (w0 + k) % nw;
      uint32_t v = __hip_atomic_load(&bits[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (v) {
        const uint32_t b = (uint32_t)__ffs((int)v) - 1u;
        const uint32_t old = atomicAnd(&bits[w], ~(1u << b));
        if (old & (1u << b)) return w * 32u + b;
        v = old & ~(1u << b);
      }
    }
    return NIL;
  }
  __device__ __forceinline__ void chunk_give(int pool, uint32_t c) const { atomicOr(&P.chunk_bits[P.chunk_word0[pool] + (c >> 5)], 1u << (c & 31u)); }
  // (one thread) make sure the query owns chunks for `need` items of a pool; false when the pool -- or the query's table -- is exhausted
  __device__ __forceinline__ bool take_chunks(uint16_t *tbl, uint32_t &owned, uint32_t need, int ch_log, int max_ch, int pool, uint32_t pool_chunks) const {
    const uint32_t want = (need + (1u << ch_log) - 1) >> ch_log;
    while (owned < want) {
      if (owned >= (uint32_t)max_ch) return false;
      const uint32_t c = P.chunk_bits ? chunk_take(pool) : atomicAdd(P.chunk_next + pool, 1u);
      if (c >= pool_chunks) return false;
      tbl[owned] = (uint16_t)c;
      owned++;
    }
    return true;
  }
  __device__ __forceinline__ bool ensure_nodes(uint32_t need) const { return take_chunks(S.node_tbl, S.node_chunks, need, NODE_CH_LOG, MAX_NODE_CH, 0, P.node_chunks); }
  __device__ __forceinline__ bool ensure_edges(uint32_t need) const { return take_chunks(S.edge_tbl, S.edge_chunks, need, EDGE_CH_LOG, MAX_EDGE_CH, 1, P.edge_chunks); }
  __device__ __forceinline__ bool ensure_open(uint32_t need) const { return take_chunks(S.open_tbl, S.open_chunks, need, OPEN_CH_LOG, MAX_OPEN_CH, 2, P.open_chunks); }
  // record field accessors
  static __device__ __forceinline__ double &g(char *r) { return *(double *)r; }
  static __device__ __forceinline__ double &h(char *r) { return *(double *)(r + 8); }
  static __device__ __forceinline__ uint32_t &flags(char *r) { return *(uint32_t *)(r + 16); }
  static __device__ __forceinline__ uint32_t &pred(char *r) { return *(uint32_t *)(r + 20); }
  static __device__ __forceinline__ int32_t *key(char *r) { return (int32_t *)(r + 24); }
  static __device__ __forceinline__ double *state(char *r) { return (double *)(r + rec_hot_bytes(CONTROL)); }
};

// thread 0: make sure the query owns chunks for `need` items of a pool; false when the pool is exhausted
__device__ __forceinline__ bool ensure_chunks(uint16_t *tbl, uint32_t &owned, uint32_t need, int ch_log, int max_ch, uint32_t *next, uint32_t pool_chunks) {
  const uint32_t want = (need + (1u << ch_log) - 1) >> ch_log;
  while (owned < want) {
    if (owned >= (uint32_t)max_ch) return false;
    uint32_t c = atomicAdd(next, 1u);
    if (c >= pool_chunks) return false;
    tbl[owned++] = (uint16_t)c;
  }
  return true;
}

// where does entry (f,g,id) belong?  -1: near set; [0,NB): fine bucket; [NB,2NB): coarse bucket
template <class SM>
__device__ __forceinline__ int classify(const SM &S, double w1, double f, double g, uint32_t id) {
  const int b1 = bucket_idx(f, S.f_base, w1);
  if (b1 != S.cur1) return b1 < S.cur1 ? -1 : NB + b1;
  const int b0 = bucket_idx(f, S.lo1, w1 * (1.0 / NB));
  if (b0 != S.cur0) return b0 < S.cur0 ? -1 : b0;
  return entry_less(f, g, id, S.ts_f, S.ts_g, S.ts_id) ? -1 : b0;
}

// link log entry idx into far bucket `code` (fine or coarse)
template <int BLOCK, int CONTROL, class SM>
__device__ __forceinline__ void far_link(const QView<BLOCK, CONTROL, SM> &Q, int code, uint32_t idx) {
  // sub-list by the bucket's own running count: the lists of a bucket differ by at most one entry, and a pull walks
  // them in lock step, one dependent HBM hop per round (by entry index the longest of 256 lists was ~3x the mean)
  const uint32_t c = atomicAdd(&Q.S.cnt[0][code], 1u);  // cnt is [2][NB]: code indexes it flat
  uint32_t old = atomicExch(&Q.bkt_head[(size_t)code * NSUB + (c & (NSUB - 1))], idx);
  Q.open(idx)->next = old;
}

// rare: the near/far boundary must drop below the active coarse bucket -> hand every fine bucket
// back to the coarse level so that the fine level can be re-bound to a lower coarse bucket
template <int BLOCK, int CONTROL, class SM>
__device__ __forceinline__ void demote_fine(const QView<BLOCK, CONTROL, SM> &Q, int tid) {
  SM &S = Q.S;
  const int c1 = NB + S.cur1;
  for (int b = 0; b < NB; b++) {
    if (S.cnt[0][b] == 0) continue;  // uniform
    uint32_t moved = 0;
    for (int sub = tid; sub < NSUB; sub += BLOCK) {
      uint32_t cur = atomicExch(&Q.bkt_head[(size_t)b * NSUB + sub], NIL);
      for (uint32_t hops = 0; cur != NIL && hops <= S.n_log; hops++) {
        const uint32_t nxt = Q.open(cur)->next;
        uint32_t old = atomicExch(&Q.bkt_head[(size_t)c1 * NSUB + (cur & (NSUB - 1))], cur);
        Q.open(cur)->next = old;
        moved++;
        cur = nxt;
      }
    }
    if (moved) atomicAdd(&S.cnt[1][S.cur1], moved);
    __syncthreads();
    if (tid == 0) S.cnt[0][b] = 0;
    __syncthreads();
  }
}

// End of a query: leave every far-bucket head of this workgroup slot empty (NIL) for the next query.
// The host clears the array once when it allocates it; a head can only be non-empty in a bucket whose
// counter is non-zero, so only those buckets are touched -- no 2 x NB x NSUB reset per query.
template <int BLOCK, int CONTROL, class SM>
__device__ __forceinline__ void clear_buckets(const QView<BLOCK, CONTROL, SM> &Q, int tid) {
  for (int b = tid; b < 2 * NB; b += BLOCK) {
    if (Q.S.cnt[0][b] == 0) continue;
    for (int sub = 0; sub < NSUB; sub++) Q.bkt_head[(size_t)b * NSUB + sub] = NIL;
  }
}

// ------------------------------------------------------------------ near-set eviction (split)
// Moves roughly the upper half of the near set (under the total order) back to the far buckets and
// lowers the near/far boundary accordingly.  Any split point keeps the structure exact.
template <int BLOCK, int CONTROL, class SM>
__device__ __forceinline__ void evict_half(const QView<BLOCK, CONTROL, SM> &Q, int tid) {
  SM &S = Q.S;
  const double w1 = Q.P.bucket_width;
  const uint32_t n = S.n_near;
  if (n < 2) return;
  for (int level = 0; level < 3; level++) {  // split on f, else on g, else on id
    double lo = INFINITY, hi = -INFINITY;
    for (uint32_t i = tid; i < n; i += BLOCK) {
      double v = level == 0 ? S.near_f[i] : level == 1 ? S.near_g[i] : (double)S.near_id[i];
      lo = fmin(lo, v);
      hi = fmax(hi, v);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      lo = fmin(lo, __shfl_xor(lo, d, 64));
      hi = fmax(hi, __shfl_xor(hi, d, 64));
    }
    if constexpr (BLOCK > 64) {
      if ((tid & 63) == 0) {
        S.red_f[tid >> 6] = lo;
        S.red_g[tid >> 6] = hi;
      }
      __syncthreads();
      lo = S.red_f[0];
      hi = S.red_g[0];
#pragma unroll
      for (int w = 1; w < BLOCK / 64; w++) {
        lo = fmin(lo, S.red_f[w]);
        hi = fmax(hi, S.red_g[w]);
      }
      __syncthreads();
    }
    if (!(lo < hi)) continue;  // all equal at this level
    if (tid < 64) S.hist[tid] = 0;
    __syncthreads();
    const double scale = 64.0 / (hi - lo);
    auto bin = [&](double v) {
      if (v >= hi) return 63;  // the maximum is always evictable (also when hi is +inf)
      double b = (v - lo) * scale;
      int bi = b >= 63.0 ? 63 : (b > 0.0 ? (int)b : 0);
      return bi;
    };
    for (uint32_t i = tid; i < n; i += BLOCK) {
      double v = level == 0 ? S.near_f[i] : level == 1 ? S.near_g[i] : (double)S.near_id[i];
      atomicAdd(&S.hist[bin(v)], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t cum = 0;
      int k = 1;
      for (int b = 0; b < 63; b++) {  // keep bins [0,k): first k with cum >= n/2, 1 <= k <= 63
        cum += S.hist[b];
        k = b + 1;
        if (cum >= n / 2) break;
      }
      S.tmp_u = (uint32_t)k;
    }
    __syncthreads();
    const int kcut = (int)S.tmp_u;
    // threshold = smallest evicted entry under the total order
    double tf = INFINITY, tg = INFINITY;
    uint32_t ti = 0xFFFFFFFFu;
    for (uint32_t i = tid; i < n; i += BLOCK) {
      double v = level == 0 ? S.near_f[i] : level == 1 ? S.near_g[i] : (double)S.near_id[i];
      if (bin(v) >= kcut && entry_less(S.near_f[i], S.near_g[i], S.near_id[i], tf, tg, ti)) {
        tf = S.near_f[i];
        tg = S.near_g[i];
        ti = S.near_id[i];
      }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      double of = __shfl_xor(tf, d, 64), og = __shfl_xor(tg, d, 64);
      uint32_t oi = __shfl_xor(ti, d, 64);
      if (entry_less(of, og, oi, tf, tg, ti)) { tf = of; tg = og; ti = oi; }
    }
    if ((tid & 63) == 0) {
      S.red_f[tid >> 6] = tf;
      S.red_g[tid >> 6] = tg;
      S.red_id[tid >> 6] = ti;
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < BLOCK / 64; w++)
      if (w == 0 || entry_less(S.red_f[w], S.red_g[w], S.red_id[w], tf, tg, ti)) { tf = S.red_f[w]; tg = S.red_g[w]; ti = S.red_id[w]; }
    const int b1t = bucket_idx(tf, S.f_base, w1);
    __syncthreads();
    if (b1t < S.cur1) {  // uniform; rare (needs an inconsistent heuristic)
      demote_fine(Q, tid);
      if (tid == 0) {
        S.cur1 = b1t;
        S.lo1 = S.f_base + (double)b1t * w1;
      }
      __syncthreads();
    }
    if (tid == 0) {
      S.cur0 = bucket_idx(tf, S.lo1, w1 * (1.0 / NB));
      S.ts_f = tf;
      S.ts_g = tg;
      S.ts_id = ti;
      S.c_evict++;
    }
    __syncthreads();
    // partition: every thread reads its strided entries, then the kept ones are re-packed
    constexpr int PER = (SM::NCAP + BLOCK - 1) / BLOCK;
    double ef[PER], eg[PER];
    uint32_t eid[PER], eix[PER];
    uint32_t keepmask = 0, nkeep = 0;
#pragma unroll
    for (int r = 0; r < PER; r++) {
      uint32_t i = tid + r * BLOCK;
      if (i < n) {
        ef[r] = S.near_f[i]; eg[r] = S.near_g[i]; eid[r] = S.near_id[i]; eix[r] = S.near_idx[i];
        const int code = classify(S, w1, ef[r], eg[r], eid[r]);
        if (code < 0) {
          keepmask |= 1u << r;
          nkeep++;
        } else {
          far_link(Q, code, eix[r]);
        }
      }
    }
    __syncthreads();
    uint32_t total;
    uint32_t base = block_excl_scan<BLOCK>(nkeep, S, tid, total);
#pragma unroll
    for (int r = 0; r < PER; r++) {
      if (keepmask & (1u << r)) {
        S.near_f[base] = ef[r]; S.near_g[base] = eg[r]; S.near_id[base] = eid[r]; S.near_idx[base] = eix[r];
        base++;
      }
    }
    if (tid == 0) S.n_near = total;
    __syncthreads();
    return;
  }
}

// lowest non-empty bucket of a level (NB if none)
template <int BLOCK, class SM>
__device__ __forceinline__ int lowest_bucket(SM &S, int level, int tid) {
  int b = NB;
  for (int i = tid; i < NB; i += BLOCK)
    if (S.cnt[level][i] > 0) { b = i; break; }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) b = min(b, __shfl_xor(b, d, 64));
  if constexpr (BLOCK > 64) {
    if ((tid & 63) == 0) S.red_id[tid >> 6] = (uint32_t)b;
    __syncthreads();
    b = (int)S.red_id[0];
#pragma unroll
    for (int w = 1; w < BLOCK / 64; w++) b = min(b, (int)S.red_id[w]);
    __syncthreads();
  }
  return b;
}

// pull every entry of far bucket `code`; entries that classify as near go to the near set, the
// rest are re-linked where they belong.  `to_near` false: activation of a coarse bucket (nothing
// is near because cur0 == -1).
template <int BLOCK, int CONTROL, class SM>
__device__ __forceinline__ void pull_bucket(const QView<BLOCK, CONTROL, SM> &Q, int code, int tid) {
  SM &S = Q.S;
  const double w1 = Q.P.bucket_width;
  uint32_t pulled = 0;
  // min(BLOCK, NSUB) sub-lists are walked at a time, one per thread: every hop is a dependent HBM
  // read, so the walk time is (longest sub-list) x (memory latency)
  constexpr int WIDE = BLOCK < NSUB ? BLOCK : NSUB;
  for (int base = 0; base < NSUB; base += WIDE) {
  uint32_t cur = NIL;
  if (tid < WIDE) cur = atomicExch(&Q.bkt_head[(size_t)code * NSUB + base + tid], NIL);
  __syncthreads();
  // (a sub-list holds log entries of this query, each at most once: more rounds than the log has entries means a list
  //  that closes on itself -- corrupted memory -- and ends the query with MPLX_PLAN_INTERNAL instead of spinning)
  const uint32_t max_rounds = S.n_log + 2u;
  for (uint32_t rounds = 0;; rounds++) {
    if (!block_any<BLOCK>(cur != NIL, S, tid)) break;
    if (rounds > max_rounds) {  // (uniform)
      if (tid == 0) {
        if (S.status < 0) S.status = 5;
        guard_mark(Q.P, GUARD_PULL, 0u, S.c_expanded, (unsigned long long)rounds);
      }
      cur = NIL;
      __syncthreads();
      break;
    }
    while (S.n_near + (uint32_t)WIDE > (uint32_t)SM::NCAP) {
      MPLX_TIC(te);
      evict_half(Q, tid);
      __syncthreads();
      MPLX_TOC(S, 3, te);
    }
    if (cur != NIL) {
      const OpenRec r = *Q.open(cur);
      const int c = classify(S, w1, r.f, r.g, r.id);
      if (c < 0) {
        uint32_t pos = atomicAdd(&S.n_near, 1u);
        S.near_f[pos] = r.f; S.near_g[pos] = r.g; S.near_id[pos] = r.id; S.near_idx[pos] = cur;
      } else {
        far_link(Q, c, cur);
      }
      pulled++;
      cur = r.next;
    }
    __syncthreads();
  }
  }
  if (pulled) atomicSub(&S.cnt[0][code], pulled);
  __syncthreads();
}

// Pull the run of fine buckets S.pull_list[0 .. S.pull_n) in ONE walk (round 6).  A sparse OPEN list -- the first thousands of
// expansions of a search, a short search altogether -- leaves a handful of entries per fine bucket, and a pull costs the same two
// dependent memory trips and half a dozen barriers whether it brings 5 entries or 500: with one bucket per pull nearly every batch
// of such a phase paid for one (C2: 36.4 ms; 31.0 ms with buckets four times as wide, which cost the deep searches 15 %; 30.6 ms with run pulls).  The
// near / far boundary is any (bucket, threshold) pair, so a refill may as well move it to the END of the last bucket of a run of
// non-empty buckets and take them all: every thread walks up to MERGE_CUR sub-lists side by side (a bucket has NSUB of them; the
// loads of a round are issued together).  The run is chosen so that it holds at most MERGE_TARGET entries and fits the near set
// next to what an iteration reserves (per-bucket counts are exact): the near set cannot overflow inside the walk, and a dense
// bucket is pulled alone, as before.  The pop order does not
// depend on any of this.
template <int BLOCK, int CONTROL, class SM>
__device__ __forceinline__ void pull_fine_run(const QView<BLOCK, CONTROL, SM> &Q, int tid) {
  SM &S = Q.S;
  const double w1 = Q.P.bucket_width;
  constexpr int MAXC = MERGE_CUR;
  const int n = S.pull_n;
  uint32_t cur[MAXC], pulled[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; c++) {
    const int li = tid + c * BLOCK, j = li / NSUB;
    pulled[c] = 0;
    cur[c] = j < n ? atomicExch(&Q.bkt_head[(size_t)S.pull_list[j] * NSUB + (li & (NSUB - 1))], NIL) : NIL;
  }
  __syncthreads();
  const uint32_t max_rounds = S.n_log + 2u;
  for (uint32_t rounds = 0;; rounds++) {
    bool any = false;
#pragma unroll
    for (int c = 0; c < MAXC; c++) any |= cur[c] != NIL;
    if (!block_any<BLOCK>(any, S, tid)) break;
    if (rounds > max_rounds) {  // (uniform) a list that closes on itself: see pull_bucket
      if (tid == 0) {
        if (S.status < 0) S.status = 5;
        guard_mark(Q.P, GUARD_PULL, 0u, S.c_expanded, (unsigned long long)rounds);
      }
      __syncthreads();
      break;
    }
    OpenRec r[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; c++)
      if (cur[c] != NIL) r[c] = *Q.open(cur[c]);
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
      if (cur[c] != NIL) {
        const int code = classify(S, w1, r[c].f, r[c].g, r[c].id);
        if (code < 0) {
          uint32_t pos = atomicAdd(&S.n_near, 1u);
          S.near_f[pos] = r[c].f; S.near_g[pos] = r[c].g; S.near_id[pos] = r[c].id; S.near_idx[pos] = cur[c];
        } else {
          far_link(Q, code, cur[c]);
        }
        pulled[c]++;
        cur[c] = r[c].next;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int c = 0; c < MAXC; c++)
    if (pulled[c]) atomicSub(&S.cnt[0][S.pull_list[(tid + c * BLOCK) / NSUB]], pulled[c]);
  __syncthreads();
}

// near set empty: bring in the lowest far entries.  false when OPEN is empty.
template <int BLOCK, int CONTROL, class SM>
__device__ __forceinline__ bool refill(const QView<BLOCK, CONTROL, SM> &Q, int tid) {
  SM &S = Q.S;
  // buckets one pull may take together: MERGE_CUR sub-list cursors per thread (none for the narrow one-node kernels)
  constexpr int MAXB = MPLX_X_MERGE_PULL && BLOCK >= NSUB ? MERGE_CUR * BLOCK / NSUB : 1;
  static_assert(MAXB <= MERGE_MAXB_ALL, "S.pull_list");
  for (int guard = 0; guard < 4 * NB; guard++) {
    const int b0 = lowest_bucket<BLOCK>(S, 0, tid);
    if (b0 < NB) {
      MPLX_TIC(t0);
      if constexpr (MAXB > 1) {
        if (tid < 64) {  // the run: non-empty buckets among the 64 from b0 on, while they hold <= target entries together
          const int b = b0 + tid;
          const uint32_t c = b < NB ? S.cnt[0][b] : 0u;
          const unsigned long long m = __ballot(c > 0u);
          const uint32_t pre = wave_incl_sum<64>(c);
          const int rank = __popcll(m & ((1ull << tid) - 1ull));
          // (what the near set can take without an eviction at the head of the next iteration: n_near + reserve <= NCAP there)
          const uint32_t used = S.n_near + S.reserve, cap = used < (uint32_t)SM::NCAP ? (uint32_t)SM::NCAP - used : 0u;
          const uint32_t target = cap < (uint32_t)MERGE_TARGET ? cap : (uint32_t)MERGE_TARGET;
          const bool take = c > 0u && (tid == 0 || (rank < MAXB && pre <= target));
          const unsigned long long tm = __ballot(take);  // (a prefix of the non-empty ones: rank and pre only grow)
          if (take) S.pull_list[rank] = b;
          if (tid == 0) {
            S.pull_n = __popcll(tm);
            S.cur0 = b0 + 63 - __clzll((long long)tm);
            S.ts_f = INFINITY;
            S.ts_g = INFINITY;
            S.ts_id = 0xFFFFFFFFu;
            S.c_refill++;
          }
        }
        __syncthreads();
        if (S.pull_n > 1) pull_fine_run(Q, tid);
        else pull_bucket(Q, b0, tid);
      } else {
        if (tid == 0) {
          S.cur0 = b0;
          S.ts_f = INFINITY;
          S.ts_g = INFINITY;
          S.ts_id = 0xFFFFFFFFu;
          S.c_refill++;
        }
        __syncthreads();
        pull_bucket(Q, b0, tid);
      }
      MPLX_TOC(S, 4, t0);
      if (S.n_near > 0) return true;
      __syncthreads();
      continue;
    }
    const int b1 = lowest_bucket<BLOCK>(S, 1, tid);
    if (b1 >= NB) return false;
    MPLX_TIC(t1);
    if (tid == 0) {  // activate coarse bucket b1: spread it over the fine level
      S.cur1 = b1;
      S.lo1 = S.f_base + (double)b1 * Q.P.bucket_width;
      S.cur0 = -1;
      S.ts_f = INFINITY;
      S.ts_g = INFINITY;
      S.ts_id = 0xFFFFFFFFu;
    }
    __syncthreads();
    pull_bucket(Q, NB + b1, tid);
    MPLX_TOC(S, 5, t1);
  }
  return false;
}

// ------------------------------------------------------------------ push one OPEN entry (log append + near/far)
template <int BLOCK, int CONTROL, class SM>
__device__ __forceinline__ void open_push(const QView<BLOCK, CONTROL, SM> &Q, uint32_t idx, double f, double g, uint32_t id) {
  SM &S = Q.S;
  if (f != f) f = INFINITY;  // never let a NaN into the order
  OpenRec *r = Q.open(idx);
  r->f = f;
  r->g = g;
  r->id = id;
  const int code = classify(S, Q.P.bucket_width, f, g, id);
  if (code < 0) {
    uint32_t pos = atomicAdd(&S.n_near, 1u);
    S.near_f[pos] = f; S.near_g[pos] = g; S.near_id[pos] = id; S.near_idx[pos] = idx;
  } else {
    far_link(Q, code, idx);
  }
}

// ------------------------------------------------------------------ commit the successors of one expansion
// `act`: this lane commits a finite-cost successor.  With all keys distinct the lanes commit in
// parallel; node / edge / log ids come from prefix sums in lane order, so they equal the ids a
// sequential loop over the control inputs would assign.
// NK: key ints (key_len_c(CONTROL), + 1 when the state's time is part of the key); lane_cost: cost of this lane's
// primitive (the voxel environment's cost depends on the control input only: P.ucost[tid]).
// YAW: the state's yaw key is the (NK + 1)-th key integer, its yaw the state double before t
// created(id, record index, L): called by the lane that has just created state `id` (its record is written)
struct NoCreateHook {
  __device__ __forceinline__ void operator()(uint32_t, uint32_t, const LaneSucc &) const {}
};
template <int BLOCK, int CONTROL, class SM, int NK = key_len_c(CONTROL), bool YAW = false, class CreateHook = NoCreateHook>
// have_v0: the caller has already loaded the first table slot of the lane's key (v0_in), e.g. while other work was in flight
__device__ __forceinline__ void commit_parallel(const QView<BLOCK, CONTROL, SM> &Q, int tid, int q, bool act, const LaneSucc &L, unsigned long long h64, double lane_cost,
                                                uint32_t action_tag, bool have_v0 = false, unsigned long long v0_in = 0, CreateHook created = CreateHook()) {
  using V = QView<BLOCK, CONTROL, SM>;
  const SearchParams &P = Q.P;
  SM &S = Q.S;
  constexpr int nk = NK, ns = key_len_c(CONTROL);
  int role = 0;  // 1 found, 2 creator
  uint32_t id = NIL;
  size_t tslot = 0;
  const unsigned long long tagq = tbl_tagq(h64, (uint32_t)q, P.tbl_epoch);
  double old_g = INFINITY, hval = 0.0;
  uint32_t fl = 0, old_pred = NIL;
  char *rec = nullptr;
  // the heuristic of a new state is needed only if the probe ends in "create", but computing it
  // for every live lane while the first table load is in flight hides ~2-3k cycles of f64 ALU
  const size_t mask = (size_t)P.table_mask;
  size_t pos = (size_t)(h64 ^ ((unsigned long long)(uint32_t)q * 0x9E3779B97F4A7C15ull)) & mask;
  unsigned long long v0 = TBL_EMPTY;
  if (act) v0 = have_v0 ? v0_in : ld_u64(&P.table[pos]);
  double hspec = 0.0;
#ifndef MPLX_NO_HSPEC
  if (act && P.eps != 0.0) {
    // get_heur is 0 when the state's key equals the goal's: with yaw the yaw key is part of that comparison
    if (YAW && L.yaw_key != S.hp.goal_yaw_key) hspec = cal_heur(S.hp, CONTROL, L.tn);
    else hspec = get_heur(S.hp, CONTROL, L.tn, L.key, nk);
  }
#endif
  if (act) {
    // MPLX_X_CLAIM_WAIT_1N (on in the product since round 5; the parity and guard suites run with it): rule R3 of DESIGN.md 3.9 for the
    // one-node kernels.  Every claim made here becomes an entry within the same expansion (there are no cut units), so a claim of
    // an EARLIER expansion with this query's tag is an entry store that has not landed yet -- wait for it instead of passing it (and
    // creating the state a second time).  The claim carries the low bits of the expansion count to tell the two apart.
    static_assert(BLOCK <= (1 << CLAIM_BATCH_SHIFT), "the thread index of a claim has nine bits");
    const bool claim_wait = MPLX_X_CLAIM_WAIT_1N || MPLX_XF(P, 64);
    const uint32_t claim_exp = claim_wait ? ((uint32_t)S.c_expanded & CLAIM_BATCH_MASK) << CLAIM_BATCH_SHIFT : 0u;
    const unsigned long long claim = tagq | (unsigned long long)(CLAIM_BASE + claim_exp + (uint32_t)tid);
    bool first = true;
    for (uint32_t steps = 0;; steps++) {
      if (steps > (1u << 22)) {  // (a probe never walks this far in a table four times the node capacity: full of foreign entries)
        S.status = 5;
        guard_mark(P, GUARD_PROBE, (uint32_t)q, S.c_expanded, (unsigned long long)pos);
        break;
      }
      unsigned long long v = first ? v0 : ld_u64(&P.table[pos]);
      first = false;
      if (tbl_empty(v, P.tbl_epoch)) {  // (cleared, or left by a batch of another epoch: claimed against the value seen)
        unsigned long long old = atomicCAS(&P.table[pos], v, claim);
        if (old == v) { role = 2; tslot = pos; break; }
        v = old;
      }
      for (uint32_t polls = 0; claim_wait && (uint32_t)v >= CLAIM_BASE && (uint32_t)v < TBL_DEAD_ID && (v & 0xFFFFFFFF00000000ull) == tagq &&
                               ((uint32_t)v & (CLAIM_BATCH_MASK << CLAIM_BATCH_SHIFT)) != claim_exp; polls++) {
        if (polls >= CLAIM_WAIT_POLLS) { S.status = 5; break; }
        if ((polls & (GUARD_POLL_EVERY - 1u)) == GUARD_POLL_EVERY - 1u) {
          guard_mark(P, GUARD_CLAIM_WAIT, (uint32_t)q, S.c_expanded, (unsigned long long)pos);
          if (guard_abort(P)) { S.status = PLAN_ABORTED; break; }
        }
        __builtin_amdgcn_s_sleep(16);
        v = ld_u64(&P.table[pos]);
      }
      const uint32_t vid = (uint32_t)v;
      if (vid < CLAIM_BASE && (v & 0xFFFFFFFF00000000ull) == tagq) {
        // one 64 B load answers: same key?  and if so g, h, flags, newest predecessor
        char *r = Q.node(vid);
        const double rg = V::g(r), rh = V::h(r);
        const uint32_t rfl = V::flags(r), rpred = V::pred(r);
        const int32_t *kk = V::key(r);
        uint32_t kd = 0;
#pragma unroll
        for (int i = 0; i < nk; i++) kd |= (uint32_t)(kk[i] ^ L.key[i]);
        if constexpr (YAW) kd |= (uint32_t)(kk[nk] ^ L.yaw_key);
        const bool eq = kd == 0u;
        if (eq) {
          role = 1; id = vid; rec = r;
          old_g = rg; hval = rh; fl = rfl; old_pred = rpred;
          break;
        }
      }
      pos = (pos + 1) & mask;
    }
  }
  uint32_t total;
  uint32_t sc = block_excl_scan<BLOCK>((role == 2 ? 1u : 0u) | (act ? 1u << 12 : 0u), S, tid, total);
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
  if (role == 2) {
    id = base_nodes + (sc & 0xFFFu);
    rec = Q.node(id);
    int32_t *kk = V::key(rec);
#pragma unroll
    for (int i = 0; i < nk; i++) kk[i] = L.key[i];
    double *st = V::state(rec);
#pragma unroll
    for (int i = 0; i < ns; i++) st[i] = i < 3 ? L.tn.p[i % 3] : i < 6 ? L.tn.v[i % 3] : i < 9 ? L.tn.a[i % 3] : L.tn.j[i % 3];
    if constexpr (YAW) {
      kk[nk] = L.yaw_key;
      st[ns] = L.yaw;
      st[ns + 1] = S.cur[0][12] + P.dt;
    } else {
      st[ns] = S.cur[0][12] + P.dt;
    }
#ifdef MPLX_NO_HSPEC
    hspec = P.eps == 0.0 ? 0.0 : get_heur(S.hp, CONTROL, L.tn, L.key, nk);
#endif
    hval = hspec;
    V::h(rec) = hval;
    st_u64(&P.table[tslot], tagq | id);
    created(id, Q.node_rec(id), L);
  }
  bool improved = false;
  double tg = 0.0;
  if (act) {
    EdgeRec *e = Q.edge(base_edges + (sc >> 12));
    e->parent = S.cur_id;
    e->next = old_pred;
    e->action = action_tag;  // control input | potential sum << EDGE_POT_SHIFT
    V::pred(rec) = base_edges + (sc >> 12);
    tg = S.cur_g + lane_cost;
    improved = tg < old_g;
    if (improved) {
      if (fl & FLAG_CLOSED) {  // re-open
        fl &= ~FLAG_CLOSED;
        atomicAdd(&S.c_reopen, 1ull);
        atomicAdd(&S.c_closed, (unsigned long long)-1ll);
      }
      fl |= FLAG_OPENED;
    }
    if (improved || role == 2) {
      V::g(rec) = improved ? tg : old_g;
      V::flags(rec) = fl;
    }
  }
  // (read ahead of the scan's barriers: thread 0 stores the new total at the end of this function, and nothing else separates
  //  that store from a slower wave's read -- see spec_commit_lanes)
  const uint32_t base_log = S.n_log;
  uint32_t total_p;
  uint32_t sp = block_excl_scan<BLOCK>(improved ? 1u : 0u, S, tid, total_p);
  if (improved) open_push(Q, base_log + sp, tg + P.eps * hval, tg, id);
  if (tid == 0) {
    S.n_nodes = base_nodes + n_new;
    S.n_edges = base_edges + n_fin;
    S.n_log = base_log + total_p;
    S.c_push += total_p;
  }
  __syncthreads();
}

// ------------------------------------------------------------------ pop the minimum valid OPEN entry
// On success S.cur_id / S.cur_g / S.cur / S.cur_key describe the node to expand and it is closed.
// EXTRA: state doubles between the control-kind's own and t (1 for yaw-carrying states: the yaw, delivered in S.cur_yaw)
template <int BLOCK, int CONTROL, class SM, int NK = key_len_c(CONTROL), int EXTRA = 0>
__device__ __forceinline__ bool pop_min(const QView<BLOCK, CONTROL, SM> &Q, int tid) {
  using V = QView<BLOCK, CONTROL, SM>;
  SM &S = Q.S;
  constexpr int nk = NK, ns = key_len_c(CONTROL);
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
    // every lane knows the winner: fetch its record (hot part + state) in one round trip
    char *rec = Q.node(bi);
    const double rg = V::g(rec);
    const uint32_t fl = V::flags(rec);
    double sval = 0.0;
    int32_t kval = 0;
    if (tid <= ns + EXTRA) sval = V::state(rec)[tid];
    if (tid < nk) kval = V::key(rec)[tid];
    if (tid == 0) {  // remove from the near set
      const uint32_t last = n - 1;
      S.near_f[bp] = S.near_f[last]; S.near_g[bp] = S.near_g[last];
      S.near_id[bp] = S.near_id[last]; S.near_idx[bp] = S.near_idx[last];
      S.n_near = last;
    }
    // stale?  (node improved since this entry was pushed, or already closed)
    const bool ok = __double_as_longlong(rg) == __double_as_longlong(bg) && !(fl & FLAG_CLOSED);
    if (ok) {
      if (tid < ns) S.cur[0][tid] = sval;
      if (EXTRA && tid == ns) S.cur_yaw[0] = sval;
      if (tid == ns + EXTRA) S.cur[0][12] = sval;
      if (tid >= ns && tid < 12) S.cur[0][tid] = 0.0;
      if (tid < nk && tid < MAX_KEY) S.cur_key[0][tid] = kval;
      if (tid == 0) {
        S.cur_id = bi;
        S.cur_g = bg;
        V::flags(rec) = fl | FLAG_CLOSED;
      }
    }
    __syncthreads();
    if (ok) return true;
  }
}

// ------------------------------------------------------------------ astar_kernel
// YAW: yaw-carrying states (use_yaw lattices, map_planner_node.cpp:119-139,165): one more key integer, one more state
// double, validate_yaw; edge costs and the heuristic are those of the yaw-less search [UNVERIFIED upstream: Jyaw is a
// trajectory metric (map_planner_node.cpp:214), not part of calculate_intrinsic_cost]
template <int BLOCK, int CONTROL, bool YAW = false>
__global__ __launch_bounds__(BLOCK) void astar_kernel(SearchParams P) {
  __shared__ Smem<BLOCK> S;
  using V = QView<BLOCK, CONTROL>;
  const int tid = threadIdx.x;
  const V Q{P, S, P.bkt_head + (size_t)blockIdx.x * 2 * NB * NSUB};
  constexpr int nk = key_len_c(CONTROL), ns = key_len_c(CONTROL), NKY = nk + (YAW ? 1 : 0), EX = YAW ? 1 : 0;
  // is_goal with the yaw tolerance of a yaw-carrying search
  auto goal_reached = [&](const State &s, double yaw, const QueryIn &in) {
    bool g = is_goal_state(s, S.hp.goal, S.hp.goal_control & 15, P.tol_pos, P.tol_vel, P.tol_acc);  // (the LDS copy of the goal)
    if (YAW && g && P.tol_yaw >= 0) g = fabs(yaw - S.hp.goal_yaw) <= P.tol_yaw;
    return g;
  };
  fill_uq<BLOCK, CONTROL>(P, S, tid);
  if ((P.xflags & 8) && blockIdx.x == 0 && tid == 0) {  // (tests: a launch that does not end by itself -- the host's deadline must)
    guard_mark(P, GUARD_TEST_HANG, 0u, 0ull, 0ull);
    while (!guard_abort(P)) __builtin_amdgcn_s_sleep(127);
  }
  for (;;) {
    if (tid == 0) {
      S.q_index = atomicAdd(P.next_query, 1);
      if (guard_abort(P)) S.q_index = P.nq;  // the host has given up on this launch: take no further query
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
      S.reserve = (uint32_t)P.n_u;
      S.node_chunks = S.edge_chunks = S.open_chunks = 0;
      S.cur1 = 0; S.cur0 = 0; S.lo1 = 0.0; S.ts_f = INFINITY; S.ts_g = INFINITY; S.ts_id = 0xFFFFFFFFu;
      S.status = -1;
      for (int i = 0; i < 10; i++) S.cyc[i] = 0;
      S.c_expanded = S.c_closed = S.c_prims = S.c_succ = S.c_succ_finite = S.c_reads = 0;
      S.c_push = S.c_reopen = S.c_refill = S.c_evict = 0;
      S.c_hash = 0;
      S.hp.w = P.w; S.hp.v_max = P.v_max; S.hp.heur_ignore_dynamics = P.heur_ignore_dynamics;
      S.hp.goal_control = in.goal_control;
      S.hp.goal = in.goal;
      S.hp.goal_nkey = state_key(in.goal_control, in.goal, S.hp.goal_key);
      S.hp.goal_yaw = in.goal_yaw;
      S.hp.goal_yaw_key = (int32_t)round(in.goal_yaw / KEY_RES_YAW);
      // PlannerBase::plan: start must be free; Astar: already at goal -> cost 0
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
      else if (in.start_t >= P.t_max || goal_reached(in.start, in.start_yaw, in)) {
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
      // ---- start node (id 0)
      if (tid == 0) {
        int32_t key[MAX_KEY + 1];
        state_key_c<CONTROL>(in.start, key);
        const int32_t ykey = (int32_t)round(in.start_yaw / KEY_RES_YAW);
        if (YAW) key[nk] = ykey;
        char *rec = Q.node(0);
        for (int i = 0; i < NKY; i++) V::key(rec)[i] = key[i];
        const double *src = (const double *)&in.start;
        for (int i = 0; i < ns; i++) V::state(rec)[i] = src[i];
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
        for (unsigned long long steps = 0;; steps++) {  // shared table: the home slot may belong to another query
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
      // ---- main loop
      for (;;) {
        while (S.n_near + S.reserve > (uint32_t)NC) {  // Smem<BLOCK>::NCAP == NC here
          MPLX_TIC(te);
          evict_half(Q, tid);
          __syncthreads();
          MPLX_TOC(S, 3, te);
        }
        MPLX_TIC(tp);
        const bool popped = pop_min<BLOCK, CONTROL, Smem<BLOCK>, NKY, EX>(Q, tid);
        MPLX_TOC(S, 0, tp);
        if (!popped) {
          if (tid == 0) S.status = 1;  // OPEN empty
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
        }
        LaneSucc L;
        MPLX_TIC(tx);
        expand_unit<BLOCK, BLOCK, CONTROL, false, true, YAW>(P, S, tid, true, L);
        MPLX_TOC(S, 1, tx);
        MPLX_TIC(tc);
        const bool act = L.valid && !L.blocked;
        {  // counters
          uint32_t tot;
          block_excl_scan<BLOCK>((L.valid ? 1u : 0u) | (act ? 1u << 10 : 0u), S, tid, tot);
          uint32_t treads;
          block_excl_scan<BLOCK>(L.reads, S, tid, treads);
          if (tid == 0) {
            S.c_prims += (unsigned long long)P.n_u;
            S.c_succ += tot & 0x3FFu;
            S.c_succ_finite += tot >> 10;
            S.c_reads += treads;
          }
        }
        // duplicate keys inside this expansion? (LDS set over the 64-bit key hashes)
        unsigned long long h64 = 0;
        S.dupset[tid] = 0;
        S.dupset[tid + BLOCK] = 0;
        __syncthreads();
        if (act) {
          if constexpr (YAW) {
            int32_t kk[MAX_KEY + 1];
#pragma unroll
            for (int i = 0; i < nk; i++) kk[i] = L.key[i];
            kk[nk] = L.yaw_key;
            h64 = key_hash64(kk, NKY);
          } else {
            h64 = key_hash64(L.key, nk);
          }
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
        // edge cost: J + w dt of the control input, plus the potential term when a potential map exists
        const double lane_cost = act ? (P.map.aux ? S.ucost_lds[tid] + P.pot_weight * (double)L.pot : S.ucost_lds[tid]) : 0.0;
        const uint32_t action_tag = (uint32_t)tid | (L.pot << EDGE_POT_SHIFT);
        if (!S.flag) {
          commit_parallel<BLOCK, CONTROL, Smem<BLOCK>, nk, YAW>(Q, tid, q, act, L, h64, lane_cost, action_tag);
        } else {
          // rare: two control inputs reach the same key -> commit one successor at a time, in order
          for (int i = 0; i < P.n_u && S.status < 0; i++) commit_parallel<BLOCK, CONTROL, Smem<BLOCK>, nk, YAW>(Q, tid, q, act && tid == i, L, h64, lane_cost, action_tag);
        }
        __syncthreads();
        MPLX_TOC(S, 2, tc);
        if (S.status >= 0) break;  // pool full
        // ---- termination tests, in the order of the reference loop: goal, max_expand (empty OPEN: next pop)
        if (tid == 0) {
          State s;
          for (int i = 0; i < 12; i++) ((double *)&s)[i] = S.cur[0][i];
          if (S.cur[0][12] >= P.t_max || goal_reached(s, S.cur_yaw[0], in))
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
        cost = S.tmp_d0;  // start already satisfied the goal
      } else if (status == 0) {
        // walk predecessor records: minimise g(pred) + edge cost, ties -> larger g(pred), then the
        // oldest record.  Written goal -> start; the host reverses.
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
            const double ec = P.map.aux ? P.ucost[er.action & EDGE_ACTION_MASK] + P.pot_weight * (double)(er.action >> EDGE_POT_SHIFT) : P.ucost[er.action & EDGE_ACTION_MASK];
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
            for (int k = 0; k < 12; k++) ts[i * 13 + k] = k < ns ? st[k] : 0.0;
            ts[i * 13 + 12] = st[ns + EX];
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
      o.spec[0] = o.spec[1] = o.spec[2] = o.spec[3] = 0;  // (one node per iteration: nothing speculative)
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

// ------------------------------------------------------------------ small utility kernels (one translation unit only)
#ifdef MPLX_UTILITY_KERNELS
// occupancy bitmap in 8x8x8 bricks; one thread per 32-bit word (4 rows of 8 voxels)
__global__ void brick_pack_kernel(const int8_t *map, int dx, int dy, int dz, int nb0, int nb1, int nb2, uint32_t *bricks) {
  const size_t nwords = (size_t)nb0 * nb1 * nb2 * 16;
  for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += (size_t)gridDim.x * blockDim.x) {
    const size_t b = w >> 4;
    const int wi = (int)(w & 15);
    const int bx = (int)(b % nb0), by = (int)((b / nb0) % nb1), bz = (int)(b / ((size_t)nb0 * nb1));
    uint32_t word = 0;
    for (int k = 0; k < 32; k++) {
      const int bit = wi * 32 + k;
      const int x = bx * 8 + (bit & 7), y = by * 8 + ((bit >> 3) & 7), z = bz * 8 + (bit >> 6);
      if (x < dx && y < dy && z < dz && map[(size_t)x + (size_t)dx * y + (size_t)dx * dy * z] > 0) word |= 1u << k;
    }
    bricks[w] = word;
  }
}

__global__ void free_unknown_kernel(int8_t *map, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride)
    if (map[i] == -1) map[i] = 0;
}

// MapUtil::dilate: every occupied voxel marks voxel + offset (inside the map) occupied, on a copy.
// Gather form: a voxel becomes occupied (100) when voxel - offset is inside and occupied.  One
// thread per voxel, x fastest (coalesced row reads; the neighbour rows come from L1/L2): 1 B read +
// 1 B written per voxel of HBM traffic.
__global__ void dilate_kernel(const int8_t *__restrict__ in, int8_t *__restrict__ out, int dx, int dy, int dz, int n_off, const int32_t *__restrict__ off) {
  const size_t n = (size_t)dx * dy * dz;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int8_t v = in[i];
    if (v <= 0) {
      const int x = (int)(i % dx), y = (int)((i / dx) % dy), z = (int)(i / ((size_t)dx * dy));
      for (int k = 0; k < n_off; k++) {
        const int sx = x - off[3 * k], sy = y - off[3 * k + 1], sz = z - off[3 * k + 2];
        if (sx < 0 || sx >= dx || sy < 0 || sy >= dy || sz < 0 || sz >= dz) continue;
        if (in[(size_t)sx + (size_t)dx * sy + (size_t)dx * dy * sz] > 0) {
          v = 100;
          break;
        }
      }
    }
    out[i] = v;
  }
}

// MapUtil::isFree/isOccupied/isUnknown/isOutside(const Veci&) for n cells
__global__ void map_cells_kernel(MapDev m, int n, const int32_t *cells, int8_t *state) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t cx = cells[3 * i], cy = cells[3 * i + 1], cz = cells[3 * i + 2];
  int8_t s = 3;
  if (cx >= 0 && cx < m.dim[0] && cy >= 0 && cy < m.dim[1] && cz >= 0 && cz < m.dim[2]) {
    int8_t v = m.data[(size_t)cx + (size_t)m.dim[0] * cy + (size_t)m.dim[0] * m.dim[1] * cz];
    s = v == 0 ? 0 : (v > 0 ? 1 : 2);
  }
  state[i] = s;
}

// MapUtil::getCloud / getFreeCloud / getUnknownCloud: voxel centres (n + 0.5) res + origin of the
// voxels of one class, in the order of the reference's loops (x outermost, z innermost).
// which: 0 occupied (> 0), 1 free (== 0), 2 unknown (< 0).  Pass 1 counts per (x, y) column.
// (which 3 / 4 are asked of the auxiliary map: voxels with a potential strictly between 0 and 100; voxels of the search region)
__device__ __forceinline__ bool cloud_match(int8_t v, int which) { return which == 0 ? v > 0 : which == 1 ? v == 0 : which == 2 ? v < 0 : which == 3 ? (v > 0 && v < 100) : v >= 0; }
__global__ void cloud_count_kernel(MapDev m, int which, uint32_t *counts) {
  const int ncol = m.dim[0] * m.dim[1];
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ncol; c += gridDim.x * blockDim.x) {
    const int x = c % m.dim[0], y = c / m.dim[0];  // consecutive threads: consecutive x (coalesced)
    uint32_t k = 0;
    for (int z = 0; z < m.dim[2]; z++) k += cloud_match(m.data[(size_t)x + (size_t)m.dim[0] * y + (size_t)m.dim[0] * m.dim[1] * z], which) ? 1u : 0u;
    counts[(size_t)x * m.dim[1] + y] = k;  // stored in output (x-major) order
  }
}
// exclusive scan of counts (one workgroup, 1024 threads, chunked with a carry); total -> *total
__global__ __launch_bounds__(1024) void cloud_scan_kernel(const uint32_t *counts, unsigned long long *offs, int n, unsigned long long *total) {
  __shared__ unsigned long long wsum[16];
  __shared__ unsigned long long carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    const unsigned long long v = i < n ? counts[i] : 0ull;
    unsigned long long x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      unsigned long long y = __shfl_up(x, d, 64);
      if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    unsigned long long pre = carry;
    for (int w = 0; w < wave; w++) pre += wsum[w];
    if (i < n) offs[i] = pre + x - v;
    __syncthreads();
    if (tid == 1023) carry = pre + x;
    __syncthreads();
  }
  if (tid == 0) *total = carry;
}
__global__ void cloud_write_kernel(MapDev m, int which, const unsigned long long *offs, unsigned long long cap, double *pts, int8_t *vals = nullptr) {
  const int ncol = m.dim[0] * m.dim[1];
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ncol; c += gridDim.x * blockDim.x) {
    const int x = c % m.dim[0], y = c / m.dim[0];
    unsigned long long o = offs[(size_t)x * m.dim[1] + y];
    for (int z = 0; z < m.dim[2]; z++) {
      if (!cloud_match(m.data[(size_t)x + (size_t)m.dim[0] * y + (size_t)m.dim[0] * m.dim[1] * z], which)) continue;
      if (o < cap) {
        pts[3 * o] = ((double)x + 0.5) * m.res + m.origin[0];
        pts[3 * o + 1] = ((double)y + 0.5) * m.res + m.origin[1];
        pts[3 * o + 2] = ((double)z + 0.5) * m.res + m.origin[2];
        if (vals) vals[o] = m.data[(size_t)x + (size_t)m.dim[0] * y + (size_t)m.dim[0] * m.dim[1] * z];
      }
      o++;
    }
  }
}

// ---- potential field / search region (MapDev::aux): 0..100 potential, < 0 outside the search region
struct PotArgs {
  double pos[3], range[3];
  int32_t n_mask;
};
// updatePotentialMap: gather form of "every occupied voxel spreads mask value H(n) to voxel + n, a voxel keeps the
// largest value"; occupied voxels hold 100; with a range only voxels whose centre lies within pos +- range get a value
__global__ void pot_update_kernel(MapDev m, int8_t *aux, PotArgs a, const int32_t *__restrict__ off, const int8_t *__restrict__ val) {
  const size_t n = (size_t)m.dim[0] * m.dim[1] * m.dim[2];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c[3] = {(int)(i % m.dim[0]), (int)((i / m.dim[0]) % m.dim[1]), (int)(i / ((size_t)m.dim[0] * m.dim[1]))};
    const bool outside_region = aux[i] < 0;
    bool in_range = true;
    for (int k = 0; k < 3; k++)
      if (a.range[k] > 0 && fabs(((double)c[k] + 0.5) * m.res + m.origin[k] - a.pos[k]) > a.range[k]) in_range = false;
    int best = 0;
    if (in_range) {
      if (m.data[i] > 0) {
        best = 100;
      } else {
        for (int k = 0; k < a.n_mask; k++) {
          const int sx = c[0] - off[3 * k], sy = c[1] - off[3 * k + 1], sz = c[2] - off[3 * k + 2];
          if (sx < 0 || sx >= m.dim[0] || sy < 0 || sy >= m.dim[1] || sz < 0 || sz >= m.dim[2]) continue;
          if (val[k] > best && m.data[(size_t)sx + (size_t)m.dim[0] * sy + (size_t)m.dim[0] * m.dim[1] * sz] > 0) best = val[k];
        }
      }
    }
    aux[i] = outside_region ? (int8_t)-1 : (int8_t)best;
  }
}
// setSearchRegion: one thread per (seed cell, offset inside +-rn): mark the voxel
__global__ void region_mark_kernel(int dx, int dy, int dz, int n_seeds, const int32_t *seeds, int rn0, int rn1, int rn2, int8_t *in) {
  const long long per = (long long)(2 * rn0 + 1) * (2 * rn1 + 1) * (2 * rn2 + 1);
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= per * n_seeds) return;
  const int s = (int)(t / per);
  long long r = t % per;
  const int ax = (int)(r % (2 * rn0 + 1)) - rn0;
  r /= (2 * rn0 + 1);
  const int ay = (int)(r % (2 * rn1 + 1)) - rn1, az = (int)(r / (2 * rn1 + 1)) - rn2;
  const int x = seeds[3 * s] + ax, y = seeds[3 * s + 1] + ay, z = seeds[3 * s + 2] + az;
  if (x < 0 || x >= dx || y < 0 || y >= dy || z < 0 || z >= dz) return;
  in[(size_t)x + (size_t)dx * y + (size_t)dx * dy * z] = 1;
}
// in == null: remove the region (hidden voxels become potential 0)
__global__ void region_apply_kernel(size_t n, const int8_t *in, int8_t *aux) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int8_t v = aux[i];
    if (!in || in[i]) { if (v < 0) aux[i] = 0; }
    else aux[i] = -1;
  }
}

__global__ void map_query_kernel(MapDev m, int n, const double *pts, int32_t *cells, int8_t *state) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int32_t c[3];
  bool out = false;
  for (int ax = 0; ax < 3; ax++) {
    c[ax] = float_to_cell(pts[3 * i + ax], m.origin[ax], m.res);
    cells[3 * i + ax] = c[ax];
    if (c[ax] < 0 || c[ax] >= m.dim[ax]) out = true;
  }
  int8_t s = 3;
  if (!out) {
    int8_t v = m.data[(size_t)c[0] + (size_t)m.dim[0] * c[1] + (size_t)m.dim[0] * m.dim[1] * c[2]];
    s = v == 0 ? 0 : (v > 0 ? 1 : 2);
  }
  state[i] = s;
}

__global__ void heuristic_kernel(SearchParams P, HeurParams hp, int n, const State *states, const double *ts, double *h, int32_t *isg) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int32_t key[MAX_KEY];
  int nk = state_key(P.control, states[i], key);
  h[i] = get_heur(hp, P.control, states[i], key, nk);
  isg[i] = (ts[i] >= P.t_max || is_goal_state(states[i], hp.goal, hp.goal_control, P.tol_pos, P.tol_vel, P.tol_acc)) ? 1 : 0;
}

#endif  // MPLX_UTILITY_KERNELS

}  // namespace mplx
