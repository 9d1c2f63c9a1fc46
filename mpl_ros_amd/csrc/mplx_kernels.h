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

// ------------------------------------------------------------------ small device helpers
__device__ __forceinline__ unsigned long long ld_u64(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_u64(unsigned long long *p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ld_u32(const uint32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool entry_less(double f1, double g1, uint32_t i1, double f2, double g2, uint32_t i2) {
  if (f1 != f2) return f1 < f2;
  if (g1 != g2) return g1 < g2;
  return i1 < i2;
}

template <int BLOCK>
struct Smem {
  // OPEN near set
  double near_f[NC], near_g[NC];
  uint32_t near_id[NC], near_idx[NC];
  uint32_t bkt_count[NB];
  // expansion scratch
  double q[18][BLOCK];  // pre-divided polynomial coefficients per primitive
  double dts[BLOCK];
  uint32_t cnt[BLOCK];  // samples per primitive (n+1), 0 if skipped
  uint32_t offs[BLOCK + 1];
  uint32_t blk[BLOCK];  // first blocked sample: (i << 1) | inside
  unsigned long long dupset[2 * BLOCK];
  double cur[13];       // state of the node being expanded (p,v,a,j,t)
  int32_t cur_key[MAX_KEY];
  HeurParams hp;
  // scan / reduce scratch
  uint32_t wsum[BLOCK / 64 + 1];
  double red_f[BLOCK / 64], red_g[BLOCK / 64];
  uint32_t red_id[BLOCK / 64], red_pos[BLOCK / 64];
  uint32_t hist[64];
  // scalars
  uint32_t n_near, n_nodes, n_edges, n_log;
  int32_t bcur;
  double ts_f, ts_g;
  uint32_t ts_id;  // split threshold inside bucket bcur
  double f_base;
  uint32_t cur_id;
  double cur_g;
  int32_t status, flag, q_index;
  uint32_t tmp_u, best_pos;
  double tmp_d0, tmp_d1;
  unsigned long long c_expanded, c_closed, c_prims, c_succ, c_succ_finite, c_reads, c_push, c_reopen, c_refill, c_evict, c_hash;
};

template <int BLOCK>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, Smem<BLOCK> &S, int tid, uint32_t &total) {
  const int lane = tid & 63, wave = tid >> 6;
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
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

template <int BLOCK>
__device__ __forceinline__ bool block_any(bool p, Smem<BLOCK> &S, int tid) {
  unsigned long long b = __ballot(p);
  if ((tid & 63) == 0) S.wsum[tid >> 6] = b != 0ull;
  __syncthreads();
  bool r = false;
#pragma unroll
  for (int w = 0; w < BLOCK / 64; w++) r = r || (S.wsum[w] != 0);
  __syncthreads();
  return r;
}

__device__ __forceinline__ int bucket_of(double f, double f_base, double width) {
  double b = floor((f - f_base) / width);
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
};

template <int BLOCK, int CONTROL>
__device__ __forceinline__ void expand_phases(const SearchParams &P, Smem<BLOCK> &S, int tid, LaneSucc &L) {
  constexpr int control = CONTROL;
  const double T = P.dt;
  L.valid = false;
  L.blocked = false;
  L.reads = 0;
  uint32_t my_cnt = 0;
  if (tid < P.n_u) {
    double c[3][6];
    const double *u = P.U + 3 * tid;
#pragma unroll
    for (int ax = 0; ax < 3; ax++) prim_build_axis(control, S.cur[ax], S.cur[3 + ax], S.cur[6 + ax], S.cur[9 + ax], u[ax], c[ax]);
#pragma unroll
    for (int ax = 0; ax < 3; ax++) {
      L.tn.p[ax] = pos_at(c[ax], T);
      L.tn.v[ax] = vel_at(c[ax], T);
      L.tn.a[ax] = acc_at(c[ax], T);
      L.tn.j[ax] = jrk_at(c[ax], T);
    }
    state_key_c<CONTROL>(L.tn, L.key);
    bool same = true;
#pragma unroll
    for (int i = 0; i < key_len_c(CONTROL); i++) same = same && (L.key[i] == S.cur_key[i]);
    double max_v;
    bool ok = !same && validate_and_maxv(control, c, T, P.v_max, P.a_max, P.j_max, &max_v);
    if (ok) {
      int n = (int)ceil(max_v * T / P.map.res);
      my_cnt = (uint32_t)(n + 1);
      S.dts[tid] = n > 0 ? T / n : 0.0;
#pragma unroll
      for (int ax = 0; ax < 3; ax++) {
        S.q[ax * 6 + 0][tid] = c[ax][0] / 120;
        S.q[ax * 6 + 1][tid] = c[ax][1] / 24;
        S.q[ax * 6 + 2][tid] = c[ax][2] / 6;
        S.q[ax * 6 + 3][tid] = c[ax][3] / 2;
        S.q[ax * 6 + 4][tid] = c[ax][4];
        S.q[ax * 6 + 5][tid] = c[ax][5];
      }
      L.valid = true;
    }
  }
  S.cnt[tid] = my_cnt;
  S.blk[tid] = 0xFFFFFFFFu;
  uint32_t total;
  uint32_t off = block_excl_scan<BLOCK>(my_cnt, S, tid, total);
  S.offs[tid] = off;
  if (tid == BLOCK - 1) S.offs[BLOCK] = total;
  __syncthreads();
  // phase 2: flattened (primitive, sample) pairs
  const int8_t *__restrict__ map = P.map.data;
  const int dx = P.map.dim[0], dy = P.map.dim[1], dz = P.map.dim[2];
  for (uint32_t e = tid; e < total; e += BLOCK) {
    int lo = 0, hi = BLOCK;  // largest p with offs[p] <= e
    while (hi - lo > 1) {
      int mid = (lo + hi) >> 1;
      if (S.offs[mid] <= e) lo = mid; else hi = mid;
    }
    const int p = lo;
    const uint32_t i = e - S.offs[p];
    const double t = (double)i * S.dts[p];
    double qq[6];
    int32_t cell[3];
#pragma unroll
    for (int ax = 0; ax < 3; ax++) {
#pragma unroll
      for (int k = 0; k < 6; k++) qq[k] = S.q[ax * 6 + k][p];
      cell[ax] = float_to_cell(pos_at_q(qq, t), P.map.origin[ax], P.map.res);
    }
    uint32_t code = 0xFFFFFFFFu;
    if (cell[0] < 0 || cell[0] >= dx || cell[1] < 0 || cell[1] >= dy || cell[2] < 0 || cell[2] >= dz) {
      code = i << 1;
    } else {
      size_t idx = (size_t)cell[0] + (size_t)dx * cell[1] + (size_t)dx * dy * cell[2];
      if (map[idx] > 0) code = (i << 1) | 1u;
    }
    if (code != 0xFFFFFFFFu) atomicMin(&S.blk[p], code);
  }
  __syncthreads();
  if (L.valid) {
    uint32_t code = S.blk[tid];
    L.blocked = code != 0xFFFFFFFFu;
    L.reads = L.blocked ? (code >> 1) + (code & 1u) : my_cnt;
  }
}

// ------------------------------------------------------------------ expand_kernel (unit-test entry)
template <int BLOCK, int CONTROL>
__global__ __launch_bounds__(BLOCK) void expand_kernel(SearchParams P, const State *nodes, const double *node_t, int K, SuccOut *out) {
  __shared__ Smem<BLOCK> S;
  const int tid = threadIdx.x;
  for (int k = blockIdx.x; k < K; k += gridDim.x) {
    if (tid < 12) S.cur[tid] = ((const double *)&nodes[k])[tid];
    if (tid == 12) S.cur[12] = node_t[k];
    __syncthreads();
    if (tid == 0) {
      State s;
      for (int i = 0; i < 12; i++) ((double *)&s)[i] = S.cur[i];
      state_key_c<CONTROL>(s, S.cur_key);
    }
    __syncthreads();
    LaneSucc L;
    expand_phases<BLOCK, CONTROL>(P, S, tid, L);
    if (tid < P.n_u) {
      SuccOut &o = out[(size_t)k * P.n_u + tid];
      for (int ax = 0; ax < 3; ax++) {
        o.pos[ax] = L.tn.p[ax];
        o.vel[ax] = L.tn.v[ax];
        o.acc[ax] = L.tn.a[ax];
        o.jrk[ax] = L.tn.j[ax];
      }
      o.yaw = 0;
      o.t = S.cur[12] + P.dt;
      o.control = P.control;
      o.enable_t = 0;
      o.cost = L.valid ? (L.blocked ? INFINITY : P.ucost[tid]) : 0.0;
      o.action = tid;
      o.valid = L.valid ? 1 : 0;
#pragma unroll
      for (int i = 0; i < 12; i++) o.key[i] = i < key_len_c(CONTROL) ? L.key[i] : 0;
      o.nkey = P.nk;
      o.voxel_reads = (int32_t)L.reads;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ per-query view of the slot pools
struct Slot {
  int32_t *node_key;
  double *node_state;
  unsigned long long *node_g;
  double *node_h;
  uint32_t *node_flags, *node_pred;
  unsigned long long *table;
  uint32_t *edge_parent, *edge_next;
  uint8_t *edge_action;
  double *log_f, *log_g;
  uint32_t *log_id, *log_next;
  uint32_t *bkt_head;
};
__device__ __forceinline__ Slot make_slot(const SearchParams &P, int s) {
  Slot L;
  size_t n = (size_t)s * P.cap_nodes, e = (size_t)s * P.cap_edges, l = (size_t)s * P.cap_log;
  L.node_key = P.node_key + n * P.nk;
  L.node_state = P.node_state + n * (P.ns + 1);
  L.node_g = P.node_g + n;
  L.node_h = P.node_h + n;
  L.node_flags = P.node_flags + n;
  L.node_pred = P.node_pred + n;
  L.table = P.table + (size_t)s * P.cap_table;
  L.edge_parent = P.edge_parent + e;
  L.edge_next = P.edge_next + e;
  L.edge_action = P.edge_action + e;
  L.log_f = P.log_f + l;
  L.log_g = P.log_g + l;
  L.log_id = P.log_id + l;
  L.log_next = P.log_next + l;
  L.bkt_head = P.bkt_head + (size_t)s * NB * NSUB;
  return L;
}

// is entry (f,g,id) in the near region?
template <int BLOCK>
__device__ __forceinline__ bool is_near(const Smem<BLOCK> &S, double width, double f, double g, uint32_t id) {
  int b = bucket_of(f, S.f_base, width);
  if (b != S.bcur) return b < S.bcur;
  return entry_less(f, g, id, S.ts_f, S.ts_g, S.ts_id);
}

// link log entry idx into its far bucket
template <int BLOCK>
__device__ __forceinline__ void far_link(Smem<BLOCK> &S, const Slot &Q, double width, double f, uint32_t idx) {
  int b = bucket_of(f, S.f_base, width);
  atomicAdd(&S.bkt_count[b], 1u);
  uint32_t old = atomicExch(&Q.bkt_head[b * NSUB + (idx & (NSUB - 1))], idx);
  Q.log_next[idx] = old;
}

// ------------------------------------------------------------------ near-set eviction (split)
// Moves roughly the upper half of the near set (under the total order) back to the far buckets and
// lowers the near/far boundary accordingly.  Any split point keeps the structure exact.
template <int BLOCK>
__device__ void evict_half(const SearchParams &P, Smem<BLOCK> &S, const Slot &Q, int tid) {
  const uint32_t n = S.n_near;
  if (n < 2) return;
  // choose the split level: f, then g, then id
  for (int level = 0; level < 3; level++) {
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
    if (!(lo < hi)) continue;  // all equal at this level (an infinite range also lands here via the bins below)
    // 64-bin histogram; bin() is monotone in v
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
      for (int b = 0; b < 63; b++) {  // keep bins [0,k): choose first k with cum >= n/2, 1 <= k <= 63
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
    if (tid == 0) {
      for (int w = 1; w < BLOCK / 64; w++)
        if (entry_less(S.red_f[w], S.red_g[w], S.red_id[w], tf, tg, ti)) { tf = S.red_f[w]; tg = S.red_g[w]; ti = S.red_id[w]; }
      S.ts_f = tf;
      S.ts_g = tg;
      S.ts_id = ti;
      S.bcur = bucket_of(tf, S.f_base, P.bucket_width);
      S.c_evict++;
    }
    __syncthreads();
    // partition: every thread reads its strided entries, then the kept ones are re-packed
    constexpr int PER = (NC + BLOCK - 1) / BLOCK;
    double ef[PER], eg[PER];
    uint32_t eid[PER], eix[PER];
    uint32_t keepmask = 0, nkeep = 0;
#pragma unroll
    for (int r = 0; r < PER; r++) {
      uint32_t i = tid + r * BLOCK;
      if (i < n) {
        ef[r] = S.near_f[i]; eg[r] = S.near_g[i]; eid[r] = S.near_id[i]; eix[r] = S.near_idx[i];
        if (entry_less(ef[r], eg[r], eid[r], S.ts_f, S.ts_g, S.ts_id)) {
          keepmask |= 1u << r;
          nkeep++;
        } else {
          far_link<BLOCK>(S, Q, P.bucket_width, ef[r], eix[r]);
        }
      }
    }
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

// ------------------------------------------------------------------ refill the near set from the lowest far bucket
template <int BLOCK>
__device__ bool refill(const SearchParams &P, Smem<BLOCK> &S, const Slot &Q, int tid) {
  int b = NB;
  for (int i = tid; i < NB; i += BLOCK)
    if (S.bkt_count[i] > 0) { b = i; break; }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) b = min(b, __shfl_xor(b, d, 64));
  if ((tid & 63) == 0) S.red_id[tid >> 6] = (uint32_t)b;
  __syncthreads();
  b = (int)S.red_id[0];
#pragma unroll
  for (int w = 1; w < BLOCK / 64; w++) b = min(b, (int)S.red_id[w]);
  __syncthreads();
  if (b >= NB) return false;
  if (tid == 0) {
    S.bcur = b;
    S.ts_f = INFINITY;
    S.ts_g = INFINITY;
    S.ts_id = 0xFFFFFFFFu;
    S.c_refill++;
  }
  uint32_t cur = NIL;
  if (tid < NSUB) cur = atomicExch(&Q.bkt_head[b * NSUB + tid], NIL);
  uint32_t pulled = 0;
  __syncthreads();
  for (;;) {
    if (!block_any<BLOCK>(cur != NIL, S, tid)) break;
    while (S.n_near > (uint32_t)(NC - BLOCK)) {
      evict_half<BLOCK>(P, S, Q, tid);
      __syncthreads();
    }
    if (cur != NIL) {
      double f = Q.log_f[cur], g = Q.log_g[cur];
      uint32_t id = Q.log_id[cur], nxt = Q.log_next[cur];
      // entries evicted during this refill may have lowered the boundary below this entry
      if (is_near<BLOCK>(S, P.bucket_width, f, g, id)) {
        uint32_t pos = atomicAdd(&S.n_near, 1u);
        S.near_f[pos] = f; S.near_g[pos] = g; S.near_id[pos] = id; S.near_idx[pos] = cur;
        pulled++;
      } else {
        // stays far: relink (bucket count already includes it)
        uint32_t old = atomicExch(&Q.bkt_head[b * NSUB + (cur & (NSUB - 1))], cur);
        Q.log_next[cur] = old;
      }
      cur = nxt;
    }
    __syncthreads();
  }
  if (pulled) atomicSub(&S.bkt_count[b], pulled);
  const bool any_pulled = block_any<BLOCK>(pulled != 0, S, tid);
  if (!any_pulled && tid == 0 && S.ts_f == INFINITY) S.bkt_count[b] = 0;  // defensive: empty lists, stale count
  __syncthreads();
  return true;
}

// ------------------------------------------------------------------ push one OPEN entry (log append + near/far)
template <int BLOCK>
__device__ __forceinline__ void open_push(const SearchParams &P, Smem<BLOCK> &S, const Slot &Q, uint32_t idx, double f, double g, uint32_t id) {
  if (f != f) f = INFINITY;  // never let a NaN into the order
  Q.log_f[idx] = f;
  Q.log_g[idx] = g;
  Q.log_id[idx] = id;
  if (is_near<BLOCK>(S, P.bucket_width, f, g, id)) {
    uint32_t pos = atomicAdd(&S.n_near, 1u);
    S.near_f[pos] = f; S.near_g[pos] = g; S.near_id[pos] = id; S.near_idx[pos] = idx;
  } else {
    far_link<BLOCK>(S, Q, P.bucket_width, f, idx);
  }
}

// ------------------------------------------------------------------ commit the successors of one expansion
// `act`: this lane commits a finite-cost successor.  With all keys distinct the lanes commit in
// parallel; node / edge / log ids come from prefix sums in lane order, so they equal the ids a
// sequential loop over the control inputs would assign.
template <int BLOCK, int CONTROL>
__device__ void commit_parallel(const SearchParams &P, Smem<BLOCK> &S, const Slot &Q, int tid, bool act, const LaneSucc &L, unsigned long long h64) {
  constexpr int nk = key_len_c(CONTROL), ns = key_len_c(CONTROL);
  int role = 0;  // 1 found, 2 creator
  uint32_t id = NIL;
  size_t tslot = 0;
  const uint32_t tag = (uint32_t)(h64 >> 32);
  if (act) {
    const size_t mask = (size_t)P.cap_table - 1;
    size_t pos = (size_t)h64 & mask;
    const unsigned long long claim = ((unsigned long long)tag << 32) | (unsigned long long)(CLAIM_BASE + (uint32_t)tid);
    for (;;) {
      unsigned long long v = ld_u64(&Q.table[pos]);
      if (v == TBL_EMPTY) {
        unsigned long long old = atomicCAS(&Q.table[pos], TBL_EMPTY, claim);
        if (old == TBL_EMPTY) { role = 2; tslot = pos; break; }
        v = old;
      }
      uint32_t vid = (uint32_t)v;
      if (vid < CLAIM_BASE && (uint32_t)(v >> 32) == tag) {
        const int32_t *kk = Q.node_key + (size_t)vid * nk;
        bool eq = true;
#pragma unroll
        for (int i = 0; i < nk; i++) eq = eq && (kk[i] == L.key[i]);
        if (eq) { role = 1; id = vid; break; }
      }
      pos = (pos + 1) & mask;
    }
  }
  uint32_t total;
  uint32_t sc = block_excl_scan<BLOCK>((role == 2 ? 1u : 0u) | (act ? 1u << 12 : 0u), S, tid, total);
  const uint32_t n_new = total & 0xFFFu, n_fin = total >> 12;
  const uint32_t base_nodes = S.n_nodes, base_edges = S.n_edges;
  const bool full = (base_nodes + n_new > P.cap_nodes) || (base_edges + n_fin > P.cap_edges);
  if (full) {
    if (tid == 0) S.status = 4;  // MPLX_PLAN_POOL_FULL
    __syncthreads();
    return;
  }
  double old_g = INFINITY, hval = 0.0;
  uint32_t fl = 0;
  if (role == 2) {
    id = base_nodes + (sc & 0xFFFu);
    int32_t *kk = Q.node_key + (size_t)id * nk;
#pragma unroll
    for (int i = 0; i < nk; i++) kk[i] = L.key[i];
    double *st = Q.node_state + (size_t)id * (ns + 1);
#pragma unroll
    for (int i = 0; i < ns; i++) st[i] = i < 3 ? L.tn.p[i % 3] : i < 6 ? L.tn.v[i % 3] : i < 9 ? L.tn.a[i % 3] : L.tn.j[i % 3];
    st[ns] = S.cur[12] + P.dt;
    hval = P.eps == 0.0 ? 0.0 : get_heur(S.hp, CONTROL, L.tn, L.key, nk);
    Q.node_h[id] = hval;
    st_u64(&Q.table[tslot], ((unsigned long long)tag << 32) | id);
  } else if (role == 1) {
    old_g = __longlong_as_double((long long)ld_u64(&Q.node_g[id]));
    fl = Q.node_flags[id];
    hval = Q.node_h[id];
  }
  bool improved = false;
  double tg = 0.0;
  if (act) {
    const uint32_t e = base_edges + (sc >> 12);
    Q.edge_parent[e] = S.cur_id;
    Q.edge_action[e] = (uint8_t)tid;
    Q.edge_next[e] = role == 2 ? NIL : Q.node_pred[id];
    Q.node_pred[id] = e;
    tg = S.cur_g + P.ucost[tid];
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
      st_u64(&Q.node_g[id], (unsigned long long)__double_as_longlong(improved ? tg : old_g));
      Q.node_flags[id] = fl;
    }
  }
  uint32_t total_p;
  uint32_t sp = block_excl_scan<BLOCK>(improved ? 1u : 0u, S, tid, total_p);
  const uint32_t base_log = S.n_log;
  if (base_log + total_p > P.cap_log) {
    if (tid == 0) S.status = 4;
    __syncthreads();
    return;
  }
  if (improved) open_push<BLOCK>(P, S, Q, base_log + sp, tg + P.eps * hval, tg, id);
  if (tid == 0) {
    S.n_nodes = base_nodes + n_new;
    S.n_edges = base_edges + n_fin;
    S.n_log = base_log + total_p;
    S.c_push += total_p;
  }
  __syncthreads();
}

// ------------------------------------------------------------------ pop the minimum valid OPEN entry
template <int BLOCK>
__device__ bool pop_min(const SearchParams &P, Smem<BLOCK> &S, const Slot &Q, int tid) {
  for (;;) {
    if (S.n_near == 0) {
      __syncthreads();
      if (!refill<BLOCK>(P, S, Q, tid)) return false;
      if (S.n_near == 0) continue;  // the pulled bucket only held entries that were relinked
    }
    const uint32_t n = S.n_near;
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
      // remove from the near set
      const uint32_t last = n - 1;
      S.near_f[bp] = S.near_f[last]; S.near_g[bp] = S.near_g[last];
      S.near_id[bp] = S.near_id[last]; S.near_idx[bp] = S.near_idx[last];
      S.n_near = last;
      // stale?  (node improved since this entry was pushed, or already closed)
      unsigned long long gb = ld_u64(&Q.node_g[bi]);
      uint32_t fl = Q.node_flags[bi];
      bool ok = gb == (unsigned long long)__double_as_longlong(bg) && !(fl & FLAG_CLOSED);
      S.flag = ok ? 1 : 0;
      if (ok) {
        S.cur_id = bi;
        S.cur_g = bg;
        Q.node_flags[bi] = fl | FLAG_CLOSED;
      }
    }
    __syncthreads();
    if (S.flag) return true;
  }
}

// ------------------------------------------------------------------ astar_kernel
template <int BLOCK, int CONTROL>
__global__ __launch_bounds__(BLOCK) void astar_kernel(SearchParams P) {
  __shared__ Smem<BLOCK> S;
  const int tid = threadIdx.x;
  const Slot Q = make_slot(P, blockIdx.x);
  constexpr int nk = key_len_c(CONTROL), ns = key_len_c(CONTROL);
  for (;;) {
    if (tid == 0) S.q_index = atomicAdd(P.next_query, 1);
    __syncthreads();
    const int q = S.q_index;
    if (q >= P.nq) break;
    const QueryIn &in = P.queries[q];
    // ---- reset the slot
    for (size_t i = tid; i < (size_t)P.cap_table; i += BLOCK) Q.table[i] = TBL_EMPTY;
    for (int i = tid; i < NB * NSUB; i += BLOCK) Q.bkt_head[i] = NIL;
    for (int i = tid; i < NB; i += BLOCK) S.bkt_count[i] = 0;
    if (tid == 0) {
      S.n_near = 0; S.n_nodes = 0; S.n_edges = 0; S.n_log = 0;
      S.bcur = 0; S.ts_f = INFINITY; S.ts_g = INFINITY; S.ts_id = 0xFFFFFFFFu;
      S.status = -1;
      S.c_expanded = S.c_closed = S.c_prims = S.c_succ = S.c_succ_finite = S.c_reads = 0;
      S.c_push = S.c_reopen = S.c_refill = S.c_evict = 0;
      S.c_hash = 0;
      S.hp.w = P.w; S.hp.v_max = P.v_max; S.hp.heur_ignore_dynamics = P.heur_ignore_dynamics;
      S.hp.goal_control = in.goal_control;
      S.hp.goal = in.goal;
      S.hp.goal_nkey = state_key(in.goal_control, in.goal, S.hp.goal_key);
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
      else if (in.start_t >= P.t_max || is_goal_state(in.start, in.goal, in.goal_control, P.tol_pos, P.tol_vel, P.tol_acc)) {
        S.status = 0;
        cost0 = 0.0;
      }
      S.tmp_d0 = cost0;
    }
    __syncthreads();
    uint32_t goal_id = NIL;
    if (S.status < 0) {
      // ---- start node (id 0)
      if (tid == 0) {
        int32_t key[MAX_KEY];
        state_key_c<CONTROL>(in.start, key);
        for (int i = 0; i < nk; i++) Q.node_key[i] = key[i];
        const double *src = (const double *)&in.start;
        for (int i = 0; i < ns; i++) Q.node_state[i] = src[i];
        Q.node_state[ns] = in.start_t;
        double h = P.eps == 0.0 ? 0.0 : get_heur(S.hp, CONTROL, in.start, key, nk);
        Q.node_h[0] = h;
        st_u64(&Q.node_g[0], (unsigned long long)__double_as_longlong(0.0));
        Q.node_flags[0] = FLAG_OPENED;
        Q.node_pred[0] = NIL;
        unsigned long long h64 = key_hash64(key, nk);
        st_u64(&Q.table[(size_t)h64 & ((size_t)P.cap_table - 1)], (h64 & 0xFFFFFFFF00000000ull) | 0ull);
        S.n_nodes = 1;
        S.f_base = 0.0 + P.eps * h;
        S.n_log = 1;
        S.c_push = 1;
      }
      __syncthreads();
      if (tid == 0) open_push<BLOCK>(P, S, Q, 0u, S.f_base, 0.0, 0u);
      __syncthreads();
      // ---- main loop
      for (;;) {
        while (S.n_near > (uint32_t)(NC - BLOCK)) {
          evict_half<BLOCK>(P, S, Q, tid);
          __syncthreads();
        }
        if (!pop_min<BLOCK>(P, S, Q, tid)) {
          if (tid == 0) S.status = 1;  // OPEN empty
          __syncthreads();
          break;
        }
        const uint32_t cur = S.cur_id;
        if (tid <= ns) S.cur[tid < ns ? tid : 12] = Q.node_state[(size_t)cur * (ns + 1) + tid];
        if (tid >= ns && tid < 12) S.cur[tid] = 0.0;
        if (tid < nk) S.cur_key[tid] = Q.node_key[(size_t)cur * nk + tid];
        if (tid == 0) {
          S.c_expanded++;
          S.c_closed++;
          S.c_hash = S.c_hash * 0x100000001B3ull + (unsigned long long)(cur + 1u);
          if (P.rec_ids && S.c_expanded <= P.cap_rec) P.rec_ids[(size_t)q * P.cap_rec + (S.c_expanded - 1)] = (int32_t)cur;
          S.flag = 0;
        }
        __syncthreads();
        LaneSucc L;
        expand_phases<BLOCK, CONTROL>(P, S, tid, L);
        const bool act = L.valid && !L.blocked;
        // counters
        {
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
          h64 = key_hash64(L.key, nk);
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
          commit_parallel<BLOCK, CONTROL>(P, S, Q, tid, act, L, h64);
        } else {
          // rare: two control inputs reach the same key -> commit one successor at a time, in order
          for (int i = 0; i < P.n_u && S.status < 0; i++) commit_parallel<BLOCK, CONTROL>(P, S, Q, tid, act && tid == i, L, h64);
        }
        __syncthreads();
        if (S.status >= 0) break;  // pool full
        // ---- termination tests, in the order of the reference loop: goal, max_expand (empty OPEN: next pop)
        if (tid == 0) {
          State s;
          for (int i = 0; i < 12; i++) ((double *)&s)[i] = S.cur[i];
          if (S.cur[12] >= P.t_max || is_goal_state(s, in.goal, in.goal_control, P.tol_pos, P.tol_vel, P.tol_acc))
            S.status = 0;
          else if (P.max_expand > 0 && S.c_expanded >= (unsigned long long)P.max_expand)
            S.status = 3;
        }
        __syncthreads();
        if (S.status >= 0) break;
      }
      goal_id = S.cur_id;
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
        bool ok = true;
        while (Q.node_pred[node] != NIL) {
          uint32_t best = NIL;
          double min_rhs = INFINITY, min_g = INFINITY;
          for (uint32_t e = Q.node_pred[node]; e != NIL; e = Q.edge_next[e]) {
            double gp = __longlong_as_double((long long)ld_u64(&Q.node_g[Q.edge_parent[e]]));
            double rhs = gp + P.ucost[Q.edge_action[e]];
            if (rhs < min_rhs || (rhs == min_rhs && gp >= min_g)) { min_rhs = rhs; min_g = gp; best = e; }
          }
          if (best == NIL || len >= MAX_TRAJ) { ok = false; break; }
          ta[len] = (int32_t)Q.edge_action[best];
          node = Q.edge_parent[best];
          len++;
          tn[len] = (int32_t)node;
          if (node == 0u) break;
        }
        if (ok) {
          cost = __longlong_as_double((long long)ld_u64(&Q.node_g[goal_id]));
          for (int i = 0; i <= len; i++) {
            const double *st = Q.node_state + (size_t)(uint32_t)tn[i] * (ns + 1);
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
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ small utility kernels
__global__ void free_unknown_kernel(int8_t *map, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride)
    if (map[i] == -1) map[i] = 0;
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

}  // namespace mplx
