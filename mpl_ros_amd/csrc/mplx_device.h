// mplx_device.h -- shared POD layouts between the host API (mplx_api.hip) and the kernels.
//
// HBM layout of the search state (all queries of a batch share three chunked pools and one hash table):
//   node pool  : records of rec_bytes(control) bytes, 32768 per chunk.  Record = hot part
//                {g f64, h f64, flags u32, pred u32, key int32[nk]} in the first 64/80 B, then the
//                first-arrival state (ns doubles) and t.  One 64 B load answers "same key? g? h?".
//   edge pool  : predecessor records {parent u32, next u32, action u32}, 65536 per chunk.
//   open pool  : OPEN-log records {f f64, g f64, id u32, next u32}, 32768 per chunk.
//   hash table : 64-bit slots {tag16 | query16 | node id32}, open addressing, shared by all queries
//                of the batch (the query index is part of the slot), cleared once per batch.
// A query owns chunks through three small chunk tables that live in LDS while it runs; chunks are
// bump-allocated from the pools and never returned within a batch.
#pragma once
#include "mplx_math.h"
#include "mplx_poly_dev.h"

namespace mplx {

constexpr int NB = 1024;          // OPEN buckets per level (coarse level 1, fine level 0)
constexpr int NSUB = 256;         // sub-lists per bucket (walked in parallel, one per thread, when a bucket is pulled)
// a refill of a sparse OPEN list pulls a run of fine buckets in one walk (mplx_kernels.h pull_fine_run)
#ifndef MPLX_X_MERGE_PULL
#define MPLX_X_MERGE_PULL 1
#endif
#ifndef MPLX_MERGE_TARGET
#define MPLX_MERGE_TARGET 512
#endif
#ifndef MPLX_MERGE_CUR
#define MPLX_MERGE_CUR 4
#endif
constexpr int MERGE_CUR = MPLX_MERGE_CUR;  // sub-list cursors per thread (8: the search kernels' register allocation suffers, tail +10 %)
constexpr int MERGE_MAXB_ALL = 16;  // buckets per run at most (512 threads x 8 cursors / NSUB)
constexpr int MERGE_TARGET = MPLX_MERGE_TARGET;  // entries per run at most (a bucket that holds more is pulled alone)
constexpr int NC = 512;           // near OPEN capacity (LDS)
constexpr int OWN = 2048;         // LDS (primitive, sample) owner map; larger expansions fall back to a search
constexpr uint32_t NIL = 0xFFFFFFFFu;
constexpr uint64_t TBL_EMPTY = 0xFFFFFFFFFFFFFFFFull;
// Slots of the batch searches' SHARED state table (astar_spec_kernel, astar_kernel, astar_poly_kernel) carry the launch epoch of the
// table (SearchParams::tbl_epoch, 0..254) in their top byte: [63:56] epoch | [55:48] tag = top byte of the key hash | [47:32] query |
// [31:0] id / claim.  A slot whose epoch is not the launch's is EMPTY -- a cleared slot (0xFF..) as well as anything an earlier batch
// left -- so the table is not cleared between batches (17-23 GB of hipMemset per C4 batch until round 5); the host clears it when
// the epoch wraps.  A claim compare-and-swaps against the stale value it saw.  (LPA*'s private tables keep the 16-bit tag and are
// cleared per fresh plan: mplx_lpa.h.)
constexpr uint32_t TBL_EPOCHS = 255;
constexpr uint32_t CLAIM_BASE = 0xFFFF0000u;  // table id field >= CLAIM_BASE: claimed in this expansion
// Speculative kernels (mplx_spec.h): a claim carries the batch it was made in -- CLAIM_BASE | (batch & 63) << 9 | thread -- and is RESOLVED
// before that batch ends: the entry of the state it was made for, or TBL_DEAD_ID when the unit that wanted the state was cut
// (the slot then stays dead: look-ups pass over it).  A look-up that meets a claim of an EARLIER batch with its query's tag is
// therefore looking at a store that has not landed yet, and waits for it; it never has to guess what a claim will become.
constexpr uint32_t TBL_DEAD_ID = 0xFFFFFFFEu;
constexpr uint32_t CLAIM_BATCH_SHIFT = 9, CLAIM_BATCH_MASK = 0x3Fu;
constexpr uint32_t CLAIM_WAIT_POLLS = 1u << 18;  // (each a sleep and a trip to memory, ~1 us: a quarter of a second, then MPLX_PLAN_INTERNAL)
constexpr uint32_t FLAG_CLOSED = 1u, FLAG_OPENED = 2u;
constexpr int MAX_TRAJ = 1024;
// predecessor record: action field = control input (low 12 bits) | potential sum of the primitive's samples << 12
// (bit 31 is the LPA* blocked flag): the edge cost is ucost[action] + pot_weight * sum, recomputed wherever it is needed
constexpr uint32_t EDGE_ACTION_MASK = 0xFFFu;
constexpr int EDGE_POT_SHIFT = 12;

constexpr int NODE_CH_LOG = 15, EDGE_CH_LOG = 16, OPEN_CH_LOG = 15;
constexpr int MAX_NODE_CH = 1024, MAX_EDGE_CH = 2048, MAX_OPEN_CH = 1536;  // per query: 33M nodes, 134M edges, 50M log
// (round 6: the OPEN log was 1024 chunks; 15 of the 1024 jerk-lattice queries of the C4 batch at the 2 M cap push 33.6 M entries and more)
constexpr int EDGE_BYTES = 12, OPEN_BYTES = 24;

constexpr int rec_hot_bytes(int control) { return control == CTRL_SNP ? 80 : 64; }
constexpr int rec_bytes(int control) { return control == CTRL_SNP ? 192 : control == CTRL_JRK ? 160 : 128; }

struct MapDev {
  const int8_t *data;      // the caller's grid, x fastest (MapUtil layout)
  const uint32_t *bricks;  // occupancy (value > 0) bit-packed in 8x8x8 bricks of 64 B: brick (bx,by,bz) at
                           // bx + nb0*(by + nb1*bz), bit (x&7) + 8*(y&7) + 64*(z&7); 1/8 of the bytes, and the
                           // <= 1-voxel steps of a primitive's samples mostly stay inside one 64 B line
  const int8_t *aux;       // potential field / search region (null: none): 0..100 potential of the voxel, < 0 outside the
                           // search region (a sample there blocks the primitive); read by the one-node kernels only
  int32_t dim[3];
  int32_t nb[3];           // bricks per axis
  double origin[3];
  double res;
};

struct QueryIn {
  State start, goal;
  double start_t;
  int32_t goal_control;
  int32_t pad;
  double start_yaw, goal_yaw;  // (yaw-carrying states only)
};

// device copy of mplx_result + diagnostics (host reads it back)
struct QueryOut {
  int32_t status, traj_len;
  double cost;
  unsigned long long n_expanded, n_closed, n_nodes, n_edges, n_primitives, n_succ, n_succ_finite, voxel_reads, n_push,
      n_reopen, n_refill, n_evict, expand_hash;
  unsigned long long t_begin, t_end;  // wall_clock64() ticks (100 MHz)
  unsigned long long cyc[10];         // s_memtime cycles: pop, expand, look-up, evict, refill, activate, commit; counts: batches, ordered batches, -
  uint32_t n_recorded, slot;
  // speculation accounting of the K-way kernels (mplx_result_speculation): OPEN entries taken as candidates, of those found stale and
  // dropped, units that ran get_succ (live candidates), units whose expansion was thrown away because the batch was cut ahead of
  // them (they return to OPEN and are expanded again later).  n_expanded counts the committed units only.
  unsigned long long spec[4];
};

// ---- helper workgroups (look-ahead expansion of a running query on otherwise idle compute units)
// get_succ(node) and the heuristic of its successors are pure functions of (node state, U, dt, limits,
// map, goal): a workgroup that has no query of its own computes them AHEAD of time for the nodes at the
// front of a running query's OPEN list and leaves the result in HBM; the leading workgroup, when it pops
// such a node, skips the voxel sampling and the heuristic root solves and only does what depends on the
// search state (table look-up, ordered commit).  Results are bit-identical by construction (same code,
// same inputs); a missing or late entry only costs time.
constexpr int WISH = 64;                       // OPEN-front entries a leader publishes per batch
// Every word a helper polls carries the launch EPOCH in its upper half: a value left over from an earlier
// launch (the two launches run on different streams; a stale cache line must never look like progress) is
// recognised as such instead of being trusted.
__host__ __device__ inline bool box_active(unsigned long long seq, uint32_t epoch) { return (uint32_t)(seq >> 32) == epoch && (uint32_t)seq != 0u; }
constexpr uint32_t CACHE_READY = 0x80000000u;  // bit 31 of CacheRec::valid (lattices have <= 31 inputs here)
constexpr int HELP_CTR_WORDS = 320;  // counters of the helper protocol: [0,32) rows + diagnostics, [32,64) done word, [128,320) debug
struct alignas(64) HelpBox {   // one per workgroup slot; every word is written by ONE 8-byte agent-scope store
  // line 0: written by the leader, polled by its helpers
  unsigned long long seq;      // epoch << 32 | n;  n = 0: no query running; 1: query running, no list yet; k + 2: list k is complete
  unsigned long long n_expanded;  // progress of the running query (helpers prefer the longest-running leader)
  uint32_t q;                  // query the leader is running
  uint32_t rank;               // position of that query in the launch order (longest predicted first)
  unsigned long long pad1[4];  // (pad1[0..3]: diagnostics of the -DMPLX_HELP_DEBUG build)
  unsigned long long xcc_plus1;  // (-DMPLX_HELP_DEBUG) XCD the leader runs on + 1 (HW_REG_XCC_ID)
  // line 1: written by helpers (atomics), read by the leader every few batches -- kept off the line the leader stores to
  uint32_t helpers;            // bit mask of attached helpers (atomicOr / atomicAnd)
  uint32_t pad2[15];
  unsigned long long wish[2][WISH];  // (q << 48) | pool index of the node record, front of OPEN first; ~0: none.
                                     // list n lives in buffer n & 1 and is announced one batch after it was written (so it is complete)
};
struct CacheRec {              // per node record of the pool; two self-validating 8-byte halves
  uint32_t row_plus1, khash;   // row of cache_h (heuristics of the successors, voxel reads in slot 31); hash of the key of
                               // the state the helper expanded -- the leader compares it with its own candidate's key
  uint32_t valid, blocked;     // per-control-input masks (valid has CACHE_READY set)
};
// Row of cache_h.  Lattices of at most 31 inputs (units of <= 64 lanes): 32 doubles -- slots 0..30 the heuristic of
// the successor of input i, slot 31 the voxel reads (as bits); the masks live in the cache record.  Larger lattices
// (units of 128 lanes, <= 128 inputs): 136 doubles -- [0..3] eight 32-bit mask words (valid x 4, blocked x 4), [4] the
// voxel reads, [8..135] the heuristics; the record's second half then only carries CACHE_READY.
// MPLX_X_ROW_PAIRS (A/B switch, off in the product -- measured in round 5: correct, 15 % slower on the blocking C4 step, profiles/r05h_*;
// host and kernels must be built alike: tools/build_variant.sh rowpairs): the small
// row as 32 pairs {value, check of that value} -- slot 2 i the heuristic of input i, 2 i + 1 its check, 62 / 63 the voxel reads and
// theirs -- each pair written by ONE 16-byte store and validated by the lane that consumes it (no cross-lane step; a torn pair
// fails whichever half is old).  DESIGN.md 7.
#ifndef MPLX_X_ROW_PAIRS
#define MPLX_X_ROW_PAIRS 0
#endif
constexpr int cache_row_doubles(int unit_lanes) { return unit_lanes <= 64 ? (MPLX_X_ROW_PAIRS ? 64 : 32) : 136; }
constexpr int cache_h_slot(int unit_lanes, int lu) { return unit_lanes <= 64 ? (MPLX_X_ROW_PAIRS ? 2 * lu : lu) : 8 + lu; }
constexpr int cache_reads_slot(int unit_lanes) { return unit_lanes <= 64 ? (MPLX_X_ROW_PAIRS ? 62 : 31) : 4; }

// ---- launch guard (round 5): no entry point may wedge its caller (the reference's plan() always returns:
// mpl_test_node/src/map_planner_node.cpp:186-196).  One block of HOST-coherent memory per context (hipHostMalloc,
// mapped): the host raises `abort` when a launch has outlived its deadline; every persistent loop of the search kernels
// reads it (system-scope load over the fabric: once per 64 batches / expansions on a wave that is off the query's serial
// chain, and inside every wait loop) and ends the query with MPLX_PLAN_ABORTED.  The kernels leave a watch record per
// workgroup slot in the same block (a posted system-scope store at the same cadence, and from a wait loop that has
// spun unusually long): what the host prints when a launch had to be aborted -- it reads its own memory, no copy
// engine, no stream, nothing that could queue behind the launch it is diagnosing.
constexpr int GUARD_SLOTS = 2048;
enum GuardPhase : uint32_t {
  GUARD_BATCH = 1,       // heartbeat of a leader: count = batches (speculative kernels) / expansions (one-node kernels)
  GUARD_CLAIM_WAIT = 2,  // look-up waiting for a claim of an earlier batch to be resolved (info: table position)
  GUARD_ROW_WAIT = 3,    // leader waiting for a look-ahead row to pass its check (info: row)
  GUARD_PROBE = 4,       // table probe that walked implausibly far (info: steps)
  GUARD_HELPER = 5,      // helper workgroup serving / looking for a leader (info: box)
  GUARD_RECOVER = 6,     // recoverTraj
  GUARD_PULL = 7,        // far-bucket walk (info: rounds)
  GUARD_DONE = 8,        // the workgroup has left the kernel's query loop
  GUARD_TEST_HANG = 9,   // the test-only spin of MPLX_X_FLAGS & 8
  GUARD_START = 10,      // query set-up (start node, table insert)
};
struct GuardRec {
  unsigned long long w0;  // phase << 56 | (query & 0xFFFF) << 40 | count (40 bits)
  unsigned long long w1;  // phase-specific
};
struct GuardBlock {
  uint32_t abort;   // 0: run; else: end every query now
  uint32_t pad[15];
  GuardRec rec[GUARD_SLOTS];
};

struct SearchParams {
  // environment
  int32_t control, n_u, ns, nk;  // ns: state doubles (without t), nk: key ints
  double dt, v_max, a_max, j_max, w, eps, tol_pos, tol_vel, tol_acc, t_max;
  int32_t max_expand, heur_ignore_dynamics;
  const double *U;      // n_u x 3
  const double *U_yaw;  // n_u yaw rates (the 4th component of Vec4f control inputs), or null
  double yaw_max, yaw_cos, tol_yaw;  // setYawmax (<= 0: no validate_yaw), cos(yaw_max) by det_sincos, yaw tolerance of is_goal (< 0: none)
  const double *ucost;  // n_u: J(control) + w dt
  MapDev map;
  double pot_weight;       // potential_weight: a free primitive costs ucost + pot_weight * (sum of the potential over its samples)
  double bucket_width;
  // shared pools
  char *node_pool, *edge_pool, *open_pool;
  uint32_t node_chunks, edge_chunks, open_chunks;  // pool sizes in chunks
  uint32_t *chunk_next;                            // [3] bump counters: node, edge, open
  // pool recycling (mplx_set_pool_recycling; round 6): a finished query hands its chunks back, so a batch's pools hold what its
  // CONCURRENT queries need, not the sum over all of them.  One bit per chunk (set: free), the three pools' words one after the
  // other; null: bump allocation from chunk_next, nothing is returned (state spaces stay readable after the batch).
  uint32_t *chunk_bits;
  uint32_t chunk_word0[3], chunk_words[3];         // first word / number of words of the node, edge, open pool
  unsigned long long *table;
  unsigned long long table_mask;                   // slots - 1
  uint32_t tbl_epoch;                              // launch epoch of the shared table's slots (tbl_empty / tbl_tagq)
  uint32_t *bkt_head;                              // per workgroup slot: 2 levels x NB x NSUB
  uint32_t cap_rec;
  // queries
  int32_t nq;
  const QueryIn *queries;
  const int32_t *order;           // launch order of the queries (longest expected first)
  QueryOut *out;
  int32_t *traj_nodes;            // nq x (MAX_TRAJ+1)
  int32_t *traj_actions;          // nq x MAX_TRAJ
  double *traj_states;            // nq x (MAX_TRAJ+1) x 13
  double *traj_yaw;               // nq x (MAX_TRAJ+1): yaw of the path states (yaw-carrying searches), or null
  int32_t *rec_ids;               // nq x cap_rec (optional)
  uint32_t *node_tables;          // nq x MAX_NODE_CH: chunk table of each query (state-space dump)
  uint32_t *edge_tables;          // nq x MAX_EDGE_CH: chunk table of the predecessor records
  int32_t *next_query;            // dynamic query counter
  // helper workgroups (null / 0 when disabled)
  HelpBox *boxes;                 // gridDim.x boxes
  CacheRec *cache_c;              // one per node-pool record, zeroed per batch
  double *cache_h;                // cache_rows x cache_row_doubles(unit lanes)
  uint32_t cache_rows;
  uint32_t *cache_next;           // bump counter of cache_h rows; [2], [3]: diagnostics
  unsigned long long *done_word;  // epoch << 32 | queries finished (helpers leave when the count reaches nq)
  uint32_t epoch;                 // launch counter of the context
  int32_t help_lead;              // workgroups blockIdx.x < help_lead lead queries (and own the boxes [0, help_lead)); the others only help
  int32_t help_max;               // helpers per leader (0 or 2..4)
  int32_t help_limit;             // workgroups of the launch that may turn into helpers once the query queue is empty (-1: no limit);
                                  // counted in cache_next[4].  Streamed batches: the rest exit and leave their compute unit to the next batch
  int32_t xflags;                 // diagnostics (MPLX_X_FLAGS): 1 table probes at agent scope, 2 release / acquire fences around a
                                  // look-ahead cache record,
                                  // 8 (tests) workgroup 0 spins until the host aborts the launch
  GuardBlock *guard;              // launch guard (host-coherent memory; never null in a product launch)
  // moving-obstacle environment (astar_poly_kernel): the worlds and the world of each query
  PolyDev poly;
  const int32_t *poly_world;
  // expansion filter (FILTER builds of astar_spec_kernel only; null / 0 otherwise): FilterView below
  const unsigned long long *filter_table;
  unsigned long long filter_mask;
  const char *filter_pool;
  uint32_t filter_flag;
};

__host__ __device__ inline bool tbl_empty(unsigned long long v, uint32_t epoch) { return (uint32_t)(v >> 56) != epoch; }
__host__ __device__ inline unsigned long long tbl_tagq(unsigned long long h64, uint32_t q, uint32_t epoch) {
  return ((unsigned long long)epoch << 56) | ((h64 >> 56) << 48) | ((unsigned long long)(q & 0xFFFFu) << 32);
}

// Expansion filter of the FILTER builds of astar_spec_kernel (getSubStateSpace by import, mplx_lpa.h): a candidate is expanded only
// if its key is found in `table` (LPA* hashing: no query bits) and the record it names in `pool` carries one of the `flag` bits.
// (Round 5 passed the four words in the moving-obstacle look-ahead slots of SearchParams; since round 6 they have fields of their
//  own at the end of the struct -- ADVICE r5.)
struct FilterView {
  const unsigned long long *table;
  unsigned long long mask;
  const char *pool;
  uint32_t flag;
};
__host__ __device__ inline FilterView filter_view(const SearchParams &P) { return FilterView{P.filter_table, P.filter_mask, P.filter_pool, P.filter_flag}; }
inline void filter_set(SearchParams &P, const unsigned long long *table, unsigned long long mask, const char *pool, uint32_t flag) {
  P.filter_table = table;
  P.filter_mask = mask;
  P.filter_pool = pool;
  P.filter_flag = flag;
}

// one successor record produced by the expand kernel (mirrors mplx_succ)
struct SuccOut {
  double pos[3], vel[3], acc[3], jrk[3];
  double yaw, t;
  int32_t control, enable_t;
  double cost;
  int32_t action, valid;
  int32_t key[12];  // (a yaw-carrying state's yaw key follows the nkey - 1 others)
  int32_t nkey, voxel_reads;
};

}  // namespace mplx
