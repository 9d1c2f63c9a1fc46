// mplx_device.h -- shared POD layouts between the host API (mplx_api.hip) and the kernels.
#pragma once
#include "mplx_math.h"

namespace mplx {

constexpr int NB = 2048;          // far OPEN buckets per query
constexpr int NSUB = 32;          // sub-lists per bucket (parallel pull)
constexpr int NC = 1024;          // near OPEN capacity (LDS)
constexpr uint32_t NIL = 0xFFFFFFFFu;
constexpr uint64_t TBL_EMPTY = 0xFFFFFFFFFFFFFFFFull;
constexpr uint32_t CLAIM_BASE = 0xFFFF0000u;  // table id field >= CLAIM_BASE: claimed in this expansion
constexpr uint32_t FLAG_CLOSED = 1u, FLAG_OPENED = 2u;
constexpr int MAX_TRAJ = 1024;

struct MapDev {
  const int8_t *data;
  int32_t dim[3];
  double origin[3];
  double res;
};

struct QueryIn {
  State start, goal;
  double start_t;
  int32_t goal_control;
  int32_t pad;
};

// device copy of mplx_result + trajectory header (host reads it back)
struct QueryOut {
  int32_t status, traj_len;
  double cost;
  unsigned long long n_expanded, n_closed, n_nodes, n_edges, n_primitives, n_succ, n_succ_finite, voxel_reads, n_push,
      n_reopen, n_refill, n_evict, expand_hash;
  uint32_t n_recorded, pad;
};

struct SearchParams {
  // environment
  int32_t control, n_u, ns, nk;  // ns: state doubles (without t), nk: key ints
  double dt, v_max, a_max, j_max, w, eps, tol_pos, tol_vel, tol_acc, t_max;
  int32_t max_expand, heur_ignore_dynamics;
  const double *U;      // n_u x 3
  const double *ucost;  // n_u: J(control) + w dt
  MapDev map;
  double bucket_width;
  // per-slot capacities
  uint32_t cap_nodes, cap_table, cap_edges, cap_log, cap_rec;
  // per-slot pools (slot s uses [s*cap, (s+1)*cap))
  int32_t *node_key;              // cap_nodes x nk
  double *node_state;             // cap_nodes x (ns+1)   (last: t)
  unsigned long long *node_g;     // f64 bits
  double *node_h;
  uint32_t *node_flags;
  uint32_t *node_pred;            // newest predecessor edge, NIL if none
  unsigned long long *table;      // cap_table: (tag << 32) | id
  uint32_t *edge_parent, *edge_next;
  uint8_t *edge_action;
  double *log_f, *log_g;
  uint32_t *log_id, *log_next;
  uint32_t *bkt_head;             // NB x NSUB
  // queries
  int32_t nq;
  const QueryIn *queries;
  QueryOut *out;
  int32_t *traj_nodes;            // nq x (MAX_TRAJ+1)
  int32_t *traj_actions;          // nq x MAX_TRAJ
  double *traj_states;            // nq x (MAX_TRAJ+1) x 13
  int32_t *rec_ids;               // nq x cap_rec (optional)
  int32_t *next_query;            // dynamic query counter
};

// one successor record produced by the expand kernel (mirrors mplx_succ)
struct SuccOut {
  double pos[3], vel[3], acc[3], jrk[3];
  double yaw, t;
  int32_t control, enable_t;
  double cost;
  int32_t action, valid;
  int32_t key[12];
  int32_t nkey, voxel_reads;
};

}  // namespace mplx
