/*
 * mplx.h -- C-ABI of the MI355X motion-primitive search back-end (libmplx.so).
 *
 * This is the drop-in boundary under the reference's C++ planner classes: a C++ shim with the MPL
 * class names (include/mpl_shim/) or any FFI binds exactly these entry points.  Plain pointers and
 * sizes only; no torch / Eigen types.  Each entry cites the reference interface it replaces
 * (paths relative to the reference repo sikang/mpl_ros; the MPL classes themselves live in the
 * un-vendored submodule motion_primitive_library, so the citations are the in-tree call sites).
 *
 * All functions return MPLX_OK (0) or a negative MPLX_ERR_* code; mplx_last_error() gives the
 * message.  plan() outcomes (no path, start occupied, ...) are reported in mplx_result.status, not
 * as errors, mirroring `bool PlannerBase::plan()` + printf diagnostics.
 * Single-threaded use per context, like the reference (one planner object, one thread).
 */
#ifndef MPLX_H
#define MPLX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes */
#define MPLX_OK 0
#define MPLX_ERR_HIP (-1)       /* a HIP runtime call failed (no GPU, OOM, launch failure) */
#define MPLX_ERR_ARG (-2)       /* invalid argument / call order */
#define MPLX_ERR_CAPACITY (-3)  /* a device pool is too small for the request */
#define MPLX_ERR_TIMEOUT (-4)   /* a search launch outlived the context's deadline (mplx_set_deadline) and was aborted: its
                                   results are void; mplx_last_error() says where every workgroup was.  The context stays
                                   usable unless the text says it is lost (the launch did not answer the abort word). */

/* plan status (mplx_result.status) */
#define MPLX_PLAN_OK 0
#define MPLX_PLAN_NO_PATH 1         /* OPEN ran empty */
#define MPLX_PLAN_START_OCCUPIED 2  /* ENV_->is_free(start.pos) failed */
#define MPLX_PLAN_MAX_EXPAND 3      /* max_expand reached */
#define MPLX_PLAN_POOL_FULL 4       /* a shared device pool is exhausted (raise mplx_set_capacity) */
#define MPLX_PLAN_INTERNAL 5        /* 64-bit key-hash collision inside one speculative batch (never observed) */
#define MPLX_PLAN_TRAJ_TOO_LONG 6   /* goal reached, mplx_result.cost is valid, but the trajectory has more than 1024
                                       primitives (the device-side recoverTraj buffer): traj_len 0, no primitives */
#define MPLX_PLAN_ABORTED 7         /* the host aborted the launch (deadline): never handed out -- the call returns
                                       MPLX_ERR_TIMEOUT -- listed for completeness of the device-side status words */

/* Control kinds = union of use_pos|use_vel|use_acc|use_jrk bits of a Waypoint
 * (mpl_test_node/src/map_planner_node.cpp:155-171 sets the bits; Control::VEL..SNP). */
#define MPLX_VEL 1
#define MPLX_ACC 3
#define MPLX_JRK 7
#define MPLX_SNP 15
/* use_yaw bit of Waypoint::control (map_planner_node.cpp:165 start.use_yaw): OR it into mplx_config.control to search
 * over yaw-carrying states (mplx_waypoint.yaw of start / goal is then read, mplx_config.U_yaw / yaw_max / tol_yaw apply) */
#define MPLX_YAW 16

/* Waypoint<3>: search-state record (fields used in-tree: map_planner_node.cpp:155-171,
 * env_poly_map.h:63-64).  yaw is read and propagated by searches configured with MPLX_YAW only. */
typedef struct {
  double pos[3], vel[3], acc[3], jrk[3];
  double yaw, t;
  int32_t control;  /* MPLX_VEL / ACC / JRK / SNP */
  int32_t enable_t; /* must be 0 for the voxel-map environment */
} mplx_waypoint;

/* Primitive<3>: 6 coefficients per axis, p(t) = c0/120 t^5 + ... + c5
 * (planning_ros_msgs/msg/Primitive.msg:2-8, primitive_ros_utils.h:12-33). */
typedef struct {
  double c[3][6];
  double t;
  int32_t control;
  int32_t pad;
  double cyaw[6]; /* yaw channel (MPLX_YAW searches): yaw(t) = cyaw[4] t + cyaw[5]; zeros otherwise */
} mplx_primitive;

/* Planner set-up = the setter calls of PlannerBase / MapPlanner
 * (setVmax/setAmax/setJmax/setDt/setU/setTol/setEpsilon/setMaxNum/setW/setHeurIgnoreDynamics:
 *  map_planner_node.cpp:176-183, map_replanner_node.cpp:415-437, ellipsoid_planner_node.cpp:67-74). */
typedef struct {
  int32_t control;  /* control kind of the search states (start.control) */
  int32_t n_u;      /* number of control inputs */
  const double *U;  /* n_u x 3, host pointer, copied */
  double dt, v_max, a_max, j_max;
  double w;         /* setW, default 10 */
  double eps;       /* setEpsilon, default 1 */
  double tol_pos, tol_vel, tol_acc; /* setTol; < 0 disables vel / acc */
  double t_max;     /* +inf unless set */
  int32_t max_expand; /* setMaxNum; <= 0 unlimited */
  int32_t heur_ignore_dynamics;
  /* yaw-carrying searches (control | MPLX_YAW): the 4th column of the Vec4f lattice (map_planner_node.cpp:119-139),
   * setYawmax (map_planner_node.cpp:182; <= 0: no validate_yaw) and setTol's yaw tolerance (< 0: disabled) */
  const double *U_yaw; /* n_u, host pointer, copied; NULL: no yaw input (the yaw stays) */
  double yaw_max;
  double tol_yaw;
} mplx_config;

/* One successor of env_map::get_succ (vec_E<Waypoint>& succ, succ_cost, action_idx:
 * env_poly_map.h:45-47, env_cloud.h:50-52) plus what the parity tests look at. */
typedef struct {
  mplx_waypoint wp;    /* tn (t = curr.t + dt) */
  double cost;         /* J + w dt, or +inf when the primitive is blocked */
  int32_t action;      /* index into U */
  int32_t valid;       /* 0: skipped (tn == curr or validate_primitive failed) */
  int32_t key[12];     /* quantised Waypoint key of tn */
  int32_t nkey;
  int32_t voxel_reads; /* map look-ups of is_free(pr), early-out honoured */
} mplx_succ;

typedef struct {
  int32_t status;      /* MPLX_PLAN_* */
  int32_t traj_len;    /* number of primitives of the recovered trajectory */
  double cost;         /* getTrajCost(); +inf when no trajectory */
  uint64_t n_expanded; /* get_succ calls == expanded_nodes_.size() */
  uint64_t n_closed;   /* getCloseSet().size() */
  uint64_t n_nodes;    /* states reached with finite cost (upstream's hm_.size() also counts states only blocked
                          primitives reach: mplx_result_blocked) */
  uint64_t n_edges;    /* predecessor records with finite cost (n_succ - n_succ_finite more have cost inf upstream) */
  uint64_t n_primitives, n_succ, n_succ_finite;
  uint64_t voxel_reads;
  uint64_t n_push, n_reopen;
  uint64_t n_refill, n_evict; /* OPEN-structure maintenance events (diagnostics) */
  uint64_t expand_hash; /* order-dependent hash of the expanded node ids (== same search) */
} mplx_result;

typedef struct mplx_ctx mplx_ctx;

/* ---- context ---- */
int mplx_ctx_create(int device, mplx_ctx **out);
void mplx_ctx_destroy(mplx_ctx *ctx);
const char *mplx_last_error(const mplx_ctx *ctx); /* ctx may be NULL: last create error */
/* run all work of this context on an existing HIP stream (hipStream_t); NULL = own stream */
int mplx_set_stream(mplx_ctx *ctx, void *hip_stream);

/* ---- MapUtil<3> (VoxelMapUtil): setMap(ori, dim, std::vector<signed char>, res)
 *      map_planner_node.cpp:10-18; grid x-fastest idx = x + dx*y + dx*dy*z (voxel_grid.cpp:88),
 *      free 0 / occupied 100 / unknown -1 (voxel_grid.h:43-45). ---- */
int mplx_map_set(mplx_ctx *ctx, const int8_t *data, const int32_t dim[3], const double origin[3], double res);
/* adopt a grid already resident in HBM (e.g. the RCCL-broadcast replica); not copied, not owned */
int mplx_map_set_device(mplx_ctx *ctx, const void *device_ptr, const int32_t dim[3], const double origin[3], double res);
int mplx_map_free_unknown(mplx_ctx *ctx);          /* MapUtil::freeUnknown, map_planner_node.cpp:71 */
int mplx_map_get(mplx_ctx *ctx, int8_t *out);      /* MapUtil::getMap, map_planner_node.cpp:35 */
int mplx_map_info(const mplx_ctx *ctx, int32_t dim[3], double origin[3], double *res);
/* batched MapUtil::floatToInt + isFree/isOccupied/isOutside for n points (xyz interleaved);
 * cells: n x 3 int32; state: 0 free, 1 occupied, 2 unknown, 3 outside */
int mplx_map_query(mplx_ctx *ctx, int n, const double *pts, int32_t *cells, int8_t *state);

/* MapUtil::dilate(const vec_Veci& neighbours), map_planner_node.cpp:75-85: every occupied voxel marks
 * voxel + offset occupied (100) when that is inside the map; evaluated on a copy, so dilation does not
 * cascade.  offsets: n_offsets x 3 int32.  The grid set with mplx_map_set_device is modified in place. */
int mplx_map_dilate(mplx_ctx *ctx, int n_offsets, const int32_t *offsets);
/* MapUtil::isFree / isOccupied / isUnknown / isOutside(const Veci&), map_replanner_node.cpp:180,217,
 * for n cells (n x 3 int32); state: 0 free, 1 occupied, 2 unknown, 3 outside */
int mplx_map_cells(mplx_ctx *ctx, int n, const int32_t *cells, int8_t *state);
/* MapUtil::rayTrace(pt1, pt2), map_replanner_node.cpp:177,208: the distinct cells met when walking from
 * pt1 to pt2 in steps of 0.8 cell, stopping at the map border.  Geometry only (no voxel is read).
 * cells: cap x 3 int32; *n_out = number of cells of the ray (may exceed cap: call again). */
int mplx_map_raytrace(const mplx_ctx *ctx, const double p1[3], const double p2[3], int32_t *cells, int cap, int *n_out);
/* MapUtil::getCloud / getFreeCloud / getUnknownCloud, map_display.cpp:244,256,266: voxel centres
 * (n + 0.5) res + origin of the occupied (which 0) / free (1) / unknown (2) voxels, x outermost and z
 * innermost like the in-tree twin voxel_grid.cpp:18-29,205-207.  pts: cap x 3 f64 (may be NULL with
 * cap 0 to query the size); *n_out = number of voxels of the class. */
int mplx_map_cloud(mplx_ctx *ctx, int which, double *pts, uint64_t cap, uint64_t *n_out);

/* ---- potential-field cost and search region (MapPlanner::setPotentialRadius / setPotentialWeight / setGradientWeight /
 *      setPotentialMapRange / updatePotentialMap / setSearchRadius / setSearchRegion / getPotentialCloud / getSearchRegion,
 *      distance_map_planner_node.cpp:185-193,199,218-224,231).  One auxiliary int8 map per context next to the grid:
 *      0..100 = potential of the voxel, < 0 = outside the search region.  While it exists, a primitive with a sample
 *      outside the region is blocked and a free primitive costs J + w dt + potential_weight * (sum of the potential over
 *      its collision samples); ACC / JRK lattices of at most 128 inputs plan on the POT builds of the speculative kernel, the
 *      rest on the one-node kernel.  [Upstream's implementation is un-vendored: semantics
 *      P1-P3 of DESIGN.md, restated for the tests' CPU checker.] ---- */
int mplx_potential_weights(mplx_ctx *ctx, double potential_weight, double gradient_weight); /* gradient_weight must be 0 */
/* updatePotentialMap(pos, range) with setPotentialRadius(radius): every occupied voxel spreads trunc(100 (1 - d)^pow),
 * d = sqrt(sum_i (n_i res / radius_i)^2) <= 1, to the voxels at offset n (largest value wins; occupied voxels hold
 * 100); range (may be NULL / 0: whole map): only voxels whose centre lies within pos +- range get a value */
int mplx_potential_update(mplx_ctx *ctx, const double radius[3], const double pos[3], const double range[3], int32_t pow_);
/* setSearchRegion(path, dense) with setSearchRadius(radius): the voxels within +-ceil(radius_i / res) of the path's
 * cells (a sparse path is joined up with rayTrace between consecutive points); n_pts 0 removes the region */
int mplx_search_region_set(mplx_ctx *ctx, int n_pts, const double *pts, const double radius[3], int dense);
int mplx_potential_clear(mplx_ctx *ctx);            /* drop the auxiliary map: plain cost, speculative kernels again */
int mplx_aux_get(mplx_ctx *ctx, int8_t *out);       /* raw copy (all 0 when none) */
/* Host wrappers that share one context (two planners on one MapUtil) tag the auxiliary map with an id of their own, so
 * that a planner can tell whether the map on the context is the one it built.  do_set != 0 stores set_value; *current
 * (may be NULL) receives the stored tag (0: nobody's; mplx_potential_clear resets it). */
int mplx_aux_token(mplx_ctx *ctx, uint64_t set_value, int32_t do_set, uint64_t *current);
/* getPotentialCloud (which 0: voxels with 0 < potential < 100, vals = the potential) / getSearchRegion (which 1): voxel
 * centres, x outermost like getCloud; *n = number of voxels of the class (may exceed cap) */
int mplx_aux_cloud(mplx_ctx *ctx, int which, double *pts, int8_t *vals, uint64_t cap, uint64_t *n);

/* ---- planner configuration ---- */
int mplx_planner_config(mplx_ctx *ctx, const mplx_config *cfg);
/* device pools: number of queries in flight (workgroups) and the TOTAL capacities shared by all
 * queries of one batch -- states, predecessor records, OPEN-log entries (0 = keep current) */
int mplx_set_capacity(mplx_ctx *ctx, int32_t n_slots, uint64_t total_nodes, uint64_t total_edges, uint64_t total_open_log);
/* Pool recycling for BATCHES on the speculative kernels (mplx_plan_batch*, mplx_stream_*; off by default): a query that has finished
 * -- its result and trajectory are written -- hands its chunks of the three pools back, and the queries that start later take them
 * again, so the capacities of mplx_set_capacity need to cover what the batch's CONCURRENTLY running queries hold (one per compute
 * unit), not the sum over all of them.  The per-query results, trajectories and counters are what they are without it (same search,
 * bit for bit); what is given up are the state spaces of the batch's queries after the call (mplx_debug_query_records refuses).
 * A single mplx_plan never recycles.  A query that finds the pools empty ends with MPLX_PLAN_POOL_FULL as before.
 * What recycling does not shrink is the shared state table: its slots are tagged with the launch's epoch, so a finished query's entries
 * stay until the launch ends -- the table (16 slots per pool state, 2^32 at most) has to hold the states one LAUNCH creates; a batch
 * that creates more is submitted as several calls (tools/c4jrk_full_cap.py). */
int mplx_set_pool_recycling(mplx_ctx *ctx, int32_t on);
/* speculative multi-node expansion (results are identical either way): -1 auto (on when
 * n_u <= 128), 0 = sequential kernel (one node per iteration), 2 = on; 8: measurement variant (eight expansion units of one
 * wave).  [Round 4's 82 -- two 256-lane workgroups per compute unit -- measured no gain and was removed in round 6.] */
int mplx_set_speculation(mplx_ctx *ctx, int32_t mode);
/* Helper workgroups: a workgroup with no query (left) to lead expands the front of a running query's OPEN list
 * ahead of time (get_succ and successor heuristics are pure functions of the node, the map and the goal) and
 * leaves the result in HBM for the workgroup that leads the query.  Results are identical with or without them.
 * One launch, at most one workgroup per compute unit: in a batch larger than the machine the leading workgroups
 * turn into helpers as they run out of queries; a batch smaller than the machine is launched with extra
 * workgroups that help from the start (up to per_leader for every query).
 * per_leader: -1 auto (4 for lattices of at most 31 inputs, 2 for the 65..128-input jerk lattices), 0 off, 2..4 (the
 * helpers of one leader split its list by record index).  reserved: workgroups that never lead, for a batch larger than the machine:
 * they help, from the start, the queries predicted longest (earliest in the launch order = longest straight-line
 * distance); 0 none, -1 auto (helpers for 1/16 of the compute units' worth of leaders -- 16 x per_leader workgroups on
 * 256 compute units -- when the batch holds at least twice as many queries
 * as the machine has compute units and max_expand is 0 or at least 200 000, i.e. one query can outlast the rest).
 * cache_rows: rows of the heuristic cache (0 auto).  Used by the speculative kernels for lattices
 * of at most 31 inputs and for the 65..128-input jerk lattices.  The leader never waits for a helper; a helper
 * leaves when every query is done, when it finds every running leader served, or when the leader it serves
 * stops completing batches. */
int mplx_set_helpers(mplx_ctx *ctx, int32_t per_leader, int32_t reserved, uint64_t cache_rows);
/* last batch: [0] heuristic-cache rows used, [1] queries finished, [2] helpers that gave up on a leader that
 * completed no batch for ~1 s of polling (0 in a healthy run), [3] helpers that left because every running leader was served */
int mplx_helper_stats(const mplx_ctx *ctx, uint32_t stats[4]);
/* f-width of one coarse OPEN bucket; the fine level divides it by 1024.  0 = default: 3*w*dt for searches on the speculative kernels with
 * lattices of at most 64 inputs, 0.5*w*dt for their larger lattices, 8*w*dt on the one-node kernels, 64*w*dt for LPA*.  A speed knob only: the
 * pop order -- and with it every result -- does not depend on it (tests/test_gpu_scale.py). */
int mplx_set_bucket_width(mplx_ctx *ctx, double width);

/* ---- env_map::get_succ for K nodes in one launch (unit-testable kernel entry).
 *      out: K x n_u records, record [k*n_u + i] belongs to control input i. ---- */
int mplx_expand_batch(mplx_ctx *ctx, int K, const mplx_waypoint *nodes, mplx_succ *out);
/* env_base::get_heur / is_goal for n states against `goal` */
int mplx_heuristic_batch(mplx_ctx *ctx, int n, const mplx_waypoint *states, const mplx_waypoint *goal, double *h, int32_t *is_goal);

/* ---- PlannerBase::plan(start, goal) -> GraphSearch::Astar, entirely on the device
 *      (map_planner_node.cpp:187).  Keeps the query's state space on the device for the getters. ---- */
int mplx_plan(mplx_ctx *ctx, const mplx_waypoint *start, const mplx_waypoint *goal, mplx_result *out);
/* nq independent queries on the shared map, one workgroup per in-flight query */
int mplx_plan_batch(mplx_ctx *ctx, int nq, const mplx_waypoint *starts, const mplx_waypoint *goals, mplx_result *out);

/* ---- streamed batches (north star: "many independent start/goal queries ... shard one-query-per-stream"; the independence
 *      of the queries: robot_team.hpp:60-66, every robot plans on its own).  mplx_plan_batch is submit + wait.  submit()
 *      returns once the batch is launched on the context's stream; wait() blocks until it is finished and hands out the
 *      results (the getters below then answer for it).  One batch may be outstanding per context. ---- */
int mplx_plan_batch_submit(mplx_ctx *ctx, int nq, const mplx_waypoint *starts, const mplx_waypoint *goals);
int mplx_plan_batch_wait(mplx_ctx *ctx, mplx_result *out /* nq records; may be NULL */);
int mplx_plan_batch_done(mplx_ctx *ctx); /* 1: wait() will not block, 0: still running, < 0: error */
/* Several batches in flight: an mplx_stream holds `depth` lanes -- contexts of their own (HIP stream, pools, result buffers)
 * on ctx's map replica (adopted, not copied) and planner set-up.  submit() launches on a free lane (MPLX_ERR_ARG when every
 * lane is busy) and returns a ticket; wait(ticket) collects it; *lane_ctx (may be NULL) is the context that answers
 * mplx_result_traj / mplx_result_timing for that batch until the lane is submitted to again.  A query is a serial pop chain
 * on one compute unit: while the longest queries of batch n finish, the workgroups of batch n + 1 run on the rest of the
 * machine.  mplx_stream_configure: pools and helper policy of every lane (helper_limit: see mplx_set_helper_limit).
 * The lanes take ctx's planner set-up when the stream is created and follow ctx's MAP: after the map was edited (setMap,
 * dilate, a VoxelGrid hand-over) the next submit makes every lane adopt it again -- with a batch still in flight that
 * submit fails (MPLX_ERR_ARG: wait first; a running batch must not lose its map).  Destroy the stream before ctx. */
typedef struct mplx_stream mplx_stream;
int mplx_stream_create(mplx_ctx *ctx, int depth, mplx_stream **out);
void mplx_stream_destroy(mplx_stream *s);
const char *mplx_stream_last_error(const mplx_stream *s);
int mplx_stream_depth(const mplx_stream *s);
int mplx_stream_configure(mplx_stream *s, int32_t n_slots, uint64_t total_nodes, uint64_t total_edges, uint64_t total_open_log,
                          int32_t helpers_per_leader, int32_t helpers_reserved, uint64_t cache_rows, int32_t helper_limit);
int mplx_stream_submit(mplx_stream *s, int nq, const mplx_waypoint *starts, const mplx_waypoint *goals, int64_t *ticket);
int mplx_stream_done(mplx_stream *s, int64_t ticket);
int mplx_stream_wait(mplx_stream *s, int64_t ticket, mplx_result *out, mplx_ctx **lane_ctx);
/* At most `limit` workgroups of a launch stay on as helpers once its query queue is empty (-1: all of them, the default
 * of a blocking batch); the others exit, so that the next batch's workgroups get their compute units. */
int mplx_set_helper_limit(mplx_ctx *ctx, int32_t limit);
/* Launch guard.  The reference's plan() always returns (mpl_test_node/src/map_planner_node.cpp:186-196); so does every
 * entry point here that waits for a search launch (mplx_plan, mplx_plan_batch, mplx_plan_batch_wait, mplx_stream_wait,
 * mplx_poly_plan_batch, mplx_lpa_plan, mplx_lpa_sub_state_space) WHEN a deadline is set: the wait polls the stream, and a launch
 * older than `seconds` -- counted from the launch, not from the call that waits for it; one clock for the leader and helper
 * launches of the moving-obstacle planner -- is told to stop through a word in host-coherent memory that every persistent loop of
 * the search kernels reads.  Opt-in: the default (<= 0; environment MPLX_DEADLINE_S) sets no deadline -- the reference's plan() has
 * no wall-clock limit either, only max_num -- and a valid but slow search is never cut short.  The call then returns MPLX_ERR_TIMEOUT -- results void,
 * mplx_last_error() lists what each workgroup was doing -- and the context remains usable.  The lanes of an mplx_stream
 * take the parent's deadline when the stream is created. */
int mplx_set_deadline(mplx_ctx *ctx, double seconds);
/* (tests) the context's next search launch spins until the deadline aborts it */
int mplx_debug_hang_next_launch(mplx_ctx *ctx);
/* (diagnostics) raw node records {g f64, h f64, flags u32, pred u32, key i32[nk], pad to 64 B, state f64[ns], t} of query q of
 * the last batch, in node-id order; *rec_size bytes each (128 VEL/ACC, 160 JRK, 192 SNP).  MPLX_ERR_CAPACITY with the counts
 * filled in when cap_bytes is too small.  (How round 5's race was found: every state checked against its key.) */
int mplx_debug_query_records(mplx_ctx *ctx, int q, uint64_t cap_bytes, void *bytes, uint64_t *n_records, int32_t *rec_size);
/* Free the context's device pools (re-created by its next plan): hands the memory to other contexts, e.g. a stream's lanes.
 * The last batch's results and trajectories stay readable; the state-space dumps of a single plan do not. */
int mplx_release_pools(mplx_ctx *ctx);

/* ---- results of query q of the last plan / plan_batch ---- */
/* getTraj(): prs[traj_len] primitives, wps[traj_len+1] waypoints, actions[traj_len]; NULLs allowed */
int mplx_result_traj(mplx_ctx *ctx, int q, mplx_primitive *prs, mplx_waypoint *wps, int32_t *actions, int32_t *node_ids);
/* getExpandedNodes(): expansion order; needs mplx_set_record(ctx, cap) before planning.
 * ids: node ids, pos: n x 3.  Returns the number written via *n (<= cap). */
int mplx_set_record(mplx_ctx *ctx, uint32_t cap_per_query);
int mplx_result_expanded(mplx_ctx *ctx, int q, uint32_t cap, int32_t *ids, uint32_t *n);
/* state-space dump of the LAST single mplx_plan(): node coords (getCloseSet / getOpenSet are
 * filters on `closed` / `opened`), g, h.  Arrays hold `cap` entries each (NULLs allowed); the call fails with
 * MPLX_ERR_CAPACITY -- writing nothing -- when the last plan created more than `cap` states (a wrapper whose own
 * plan is no longer the context's last one must not be handed another planner's, larger, state space). */
int mplx_result_nodes(mplx_ctx *ctx, uint64_t cap, mplx_waypoint *coords, double *g, double *h, int32_t *closed, int32_t *opened);

/* StateSpace predecessor lists of the LAST single mplx_plan(): the reference keeps pred_coord /
 * pred_action_id / pred_action_cost per node (poly_map_planner.h:70-86) and getAllPrimitives()
 * (poly_map_replanner_node.cpp:184,234) walks them.  For every node in id order its predecessor edges in
 * arrival order (the order of the reference's push_back): child / parent are node ids, action indexes U (edge cost = J(U[action]) + w dt).
 * Arrays sized mplx_result.n_edges (NULLs allowed); *n = number of edges of the state space. */
int mplx_result_edges(mplx_ctx *ctx, int32_t *child, int32_t *parent, int32_t *action, uint64_t cap, uint64_t *n);

/* Blocked primitives of the LAST single mplx_plan().  The reference's GraphSearch gives EVERY successor get_succ
 * returns an hm_ entry and a pred_coord / pred_action_id / pred_action_cost entry, also the ones whose cost is +inf
 * (is_free(pr) failed; the successor is still emitted: env_poly_map.h:60-66), so upstream's getAllPrimitives()
 * contains them and hm_.size() counts the states only they reach.  The device search stores only what can be
 * relaxed (mplx_result.n_nodes / n_edges count finite arrivals); this call re-derives the rest on request with one
 * get_succ launch over the closed nodes.  parent / action: one entry per blocked primitive (parents in node-id
 * order); *n = their number (may exceed cap); *n_states_all = hm_.size() as upstream counts it. */
int mplx_result_blocked(mplx_ctx *ctx, int32_t *parent, int32_t *action, uint64_t cap, uint64_t *n, uint64_t *n_states_all);

/* ---- LPA*: incremental replanning on a state space that stays on the device between plan() calls
 *      (PlannerBase::setLPAstar(true), map_replanner_node.cpp:425-437).  An mplx_lpa is the state space of ONE such
 *      planner on ctx's map, with pools of its own -- an A* planner sharing the MapUtil / context (planner_ next to
 *      replan_planner_, map_replanner_node.cpp:415,427) does not disturb it.  It plans with the set-up last given to
 *      mplx_planner_config(ctx).  The search is Koenig & Likhachev's LPA* as upstream structures it; the un-vendored
 *      details are restated in oracle/mpl_oracle_lpa.inc (choices L1-L6), the in-tree anchor of the repair mechanism is
 *      PolyMapPlanner::updateNodes (poly_map_planner.h:61-93). ---- */
typedef struct mplx_lpa mplx_lpa;
int mplx_lpa_create(mplx_ctx *ctx, mplx_lpa **out);  /* no device work; destroy it before ctx */
void mplx_lpa_destroy(mplx_lpa *l);
const char *mplx_lpa_last_error(const mplx_lpa *l);
int mplx_lpa_set_capacity(mplx_lpa *l, uint64_t nodes, uint64_t edges, uint64_t open_log); /* per state space (0 = keep); two are held */
int mplx_lpa_set_record(mplx_lpa *l, uint32_t cap);
/* PlannerBase::plan (map_replanner_node.cpp:141): ComputeShortestPath on the kept state space when the goal, the planner
 * set-up and the start (= the current root) are those of the previous plan, else on a new one.  out->n_expanded = states
 * popped by THIS call; n_nodes / n_edges = size of the whole state space. */
int mplx_lpa_plan(mplx_lpa *l, const mplx_waypoint *start, const mplx_waypoint *goal, mplx_result *out);
int mplx_lpa_initialized(const mplx_lpa *l);         /* PlannerBase::initialized(), map_replanner_node.cpp:195,232,244 */
int mplx_lpa_reset(mplx_lpa *l);                     /* PlannerBase::reset() */
/* MapPlanner::updateBlockedNodes / updateClearedNodes(const vec_Vec3i&) (map_replanner_node.cpp:196,233), to be called
 * after the context's map was edited: every stored predecessor entry (and, for cleared, every successor that had been
 * emitted with cost +inf) is re-evaluated against the current map -- the in-tree mechanism of poly_map_planner.h:61-93 --
 * increaseCost / decreaseCost.  cells (n x 3 int32): the voxels the caller changed (nothing happens for an empty list).
 * *n_changed: predecessor entries whose cost changed (the primitives upstream returns). */
int mplx_lpa_update_blocked(mplx_lpa *l, int n_cells, const int32_t *cells, uint64_t *n_changed);
int mplx_lpa_update_cleared(mplx_lpa *l, int n_cells, const int32_t *cells, uint64_t *n_changed);
/* PlannerBase::getSubStateSpace(time_step) (map_replanner_node.cpp:245): re-root the state space at the time_step-th
 * state of the last trajectory (the caller then plans from getTraj().getWaypoints()[time_step]); also compacts the pools */
int mplx_lpa_sub_state_space(mplx_lpa *l, int32_t time_step);
/* How getSubStateSpace re-roots (results of every later plan are the same either way; the state spaces differ in which states they
 * keep).  0: Dijkstra from the new root through the states that had been expanded in the space being left (round 4/5).  1: an A*
 * from the new root to the planner's goal, kept as the new space -- the cheaper of the two as soon as the old space is large (C2
 * size: 23 ms against 113 ms).  2 (default): 1 when the space being left holds more than 16384 states, else 0. */
int mplx_lpa_set_reroot(mplx_lpa *l, int32_t mode);
/* results: the stored trajectory of the last successful plan; the state space (rhs next to g; built = expanded at least once;
 * blocked = the entry's primitive is not free in the current map) */
int mplx_lpa_traj_len(const mplx_lpa *l);
int mplx_lpa_result_traj(mplx_lpa *l, mplx_primitive *prs, mplx_waypoint *wps, int32_t *actions, int32_t *node_ids);
int mplx_lpa_counts(const mplx_lpa *l, uint64_t *n_nodes, uint64_t *n_edges, uint64_t *n_blocked_log);
int mplx_lpa_result_nodes(mplx_lpa *l, uint64_t cap, mplx_waypoint *coords, double *g, double *rhs, double *h, int32_t *closed, int32_t *opened, int32_t *built);
int mplx_lpa_result_edges(mplx_lpa *l, int32_t *child, int32_t *parent, int32_t *action, int32_t *blocked, uint64_t cap, uint64_t *n);
int mplx_lpa_result_expanded(mplx_lpa *l, uint32_t cap, int32_t *ids, uint32_t *n);
int mplx_lpa_last_kernel_ms(const mplx_lpa *l, float *ms);

/* ---- VoxelGrid (planning_ros_utils/src/mapping_utils/voxel_grid.cpp, the mapper in front of the planner:
 *      map_replanner_node.cpp:17,181,218,329-331, cloud_to_map.cpp:11-12).  Device-resident; same
 *      semantics as the in-tree class: float resolution, truncating floatToInt (:201-203), two grids
 *      (map_, inflated_map_), values free 0 / occupied 100 decaying by 1 per decay(). ---- */
typedef struct mplx_grid mplx_grid;
int mplx_grid_create(int device, const double origin[3], const double dim[3], float res, mplx_grid **out); /* ctor :3-10 */
void mplx_grid_destroy(mplx_grid *g);
const char *mplx_grid_last_error(const mplx_grid *g);
int mplx_grid_allocate(mplx_grid *g, const double new_dim_d[3], const double new_ori_d[3], int *changed);   /* :129-181 */
int mplx_grid_info(const mplx_grid *g, int32_t dim[3], double origin_d[3], float *res);
int mplx_grid_clear(mplx_grid *g);                                                                      /* :12-16 */
int mplx_grid_add_cloud(mplx_grid *g, int n, const double *pts /* n x 3 */);                           /* :183-189 */
/* addCloud(pts, ns) :191-207 -- new_obs (cap x 3): the cells of inflated_map_ that became occupied, in the
 * order the reference's sequential loop flips them; *n_new = their number (may exceed cap) */
int mplx_grid_add_cloud_inflate(mplx_grid *g, int n, const double *pts, int n_ns, const int32_t *ns, int32_t *new_obs, int cap, int *n_new);
int mplx_grid_decay(mplx_grid *g);                                                                      /* :213-224 */
int mplx_grid_clear_column(mplx_grid *g, int nx, int ny);                                               /* clear(nx, ny) :31-33 */
int mplx_grid_fill_column(mplx_grid *g, int nx, int ny);                                                /* fill(nx, ny) :35-39 */
int mplx_grid_fill_cell(mplx_grid *g, int nx, int ny, int nz);                                          /* fill(nx, ny, nz) :41-46 */
/* getMap() / getInflatedMap() :71-127: VoxelMap.data (x fastest), > 0 -> 100, everything else 0 */
int mplx_grid_get_map(mplx_grid *g, int inflated, int8_t *data);
int mplx_grid_get_cloud(mplx_grid *g, double *pts, uint64_t cap, uint64_t *n);                         /* getCloud() :18-29 */
/* getMap() handed to a planner context device to device (= setMap(map_util, voxel_mapper_->getMap()),
 * map_replanner_node.cpp:186-188,224-226, without the host round trip) */
int mplx_grid_to_map(mplx_grid *g, int inflated, mplx_ctx *ctx);

/* ---- moving-obstacle environment: env_poly_map / PolyMapUtil / collide() of
 *      mpl_external_planner/include/mpl_external_planner/poly_map_planner/ (env_poly_map.h:45-73, poly_map_util.h:72-109,
 *      primitive_geometry_utils.h:5-173, simple_obstacle.h), 2-D like the multi-robot node.  One object holds several
 *      WORLDS -- what one robot's planner sees: bounding box, start time, static / linear / nonlinear obstacles -- so
 *      that the 16 planners of a decentralised tick (robot_team.hpp:33-66) run in one launch.  get_succ: any control kind
 *      (poly_map_planner_node.cpp:73-85 exposes use_acc / use_jrk) and obstacle trajectories of any degree <= 5, through the
 *      general solve(a, b, c, d, e, f) of primitive_geometry_utils.h:28,71,148; the search: ACC or JRK states. ---- */
typedef struct mplx_poly mplx_poly;
typedef struct {
  double state[9];     /* tn: pos2 vel2 acc2 jrk2, t = curr.t + dt (enable_t, env_poly_map.h:63-64) */
  double cost;         /* calculate_intrinsic_cost(pr) = J(control) + 0.001 J(VEL) + w dt, or +inf when isFree(pr, t) fails */
  int32_t action;      /* index into U */
  int32_t valid;       /* 0: skipped (end point outside the bounding box, or validate_primitive failed) */
} mplx_poly_succ;
int mplx_poly_create(int device, mplx_poly **out);
void mplx_poly_destroy(mplx_poly *p);
const char *mplx_poly_last_error(const mplx_poly *p);
/* planner set-up: control kind (MPLX_VEL / ACC / JRK / SNP), control inputs U (n_u x 2), dt, limits, time weight w */
int mplx_poly_config(mplx_poly *p, int32_t control, int32_t n_u, const double *U, double dt, double v_max, double a_max, double j_max, double w);
/* (re)build the worlds: begin(n), set_world + add_* per world, commit() uploads them */
int mplx_poly_begin(mplx_poly *p, int32_t n_worlds);
int mplx_poly_set_world(mplx_poly *p, int32_t world, const double ori[2], const double dim[2], double start_t); /* setMap + setStartTime, poly_map_planner.h:30-36 */
/* hp: n_hp x {px, py, nx, ny} = Hyperplane2D(p, n) of the obstacle's Polyhedron2D; pt: representative point */
int mplx_poly_add_static(mplx_poly *p, int32_t world, int32_t n_hp, const double *hp, const double pt[2]);                            /* PolyhedronObstacle */
int mplx_poly_add_linear(mplx_poly *p, int32_t world, int32_t n_hp, const double *hp, const double pt[2], const double v[2], double cov_v); /* PolyhedronLinearObstacle */
/* segs: n_seg x {cx[6], cy[6], T} primitives of the obstacle's trajectory; start_t, disappear_front/back: simple_obstacle.h:122-160 */
int mplx_poly_add_nonlinear(mplx_poly *p, int32_t world, int32_t n_hp, const double *hp, int32_t n_seg, const double *segs, double start_t, int32_t dis_front, int32_t dis_back);
int mplx_poly_commit(mplx_poly *p);
/* env_poly_map::get_succ for K nodes in one launch; node k (state: pos2 vel2 acc2 jrk2 t) lives in world world_of[k].
 * out: K x n_u records, record [k * n_u + i] belongs to control input i */
int mplx_poly_get_succ_batch(mplx_poly *p, int32_t K, const int32_t *world_of, const double *states, mplx_poly_succ *out);
/* PlannerBase::plan through env_poly_map for n queries in ONE launch, one workgroup per query (the planners of a
 * decentralised tick, robot.hpp:92-133): query k plans in world world_of[k] from starts[k] (pos2 vel2 acc2 jrk2 t) to
 * goals[k] (pos2 vel2 ...).  States are keyed with their time (enable_t, env_poly_map.h:63-64).  setEpsilon / setTol /
 * setMaxNum / setHeurIgnoreDynamics are the eps / tol_* / max_expand / heur_ignore_dynamics arguments.  ACC or JRK control
 * (an SNP state keyed with its time would need 13 key integers; refused). */
int mplx_poly_set_capacity(mplx_poly *p, int32_t n_slots, uint64_t total_nodes, uint64_t total_edges, uint64_t total_open_log);
int mplx_poly_plan_batch(mplx_poly *p, int32_t n, const int32_t *world_of, const double *starts, const double *goals, double eps, double tol_pos,
                         double tol_vel, int32_t max_expand, int32_t heur_ignore_dynamics, mplx_result *out);
/* trajectory of query q of the last batch: actions[traj_len], node_ids[traj_len + 1], states (traj_len + 1) x 9; NULLs allowed */
int mplx_poly_result_traj(mplx_poly *p, int32_t q, int32_t *actions, int32_t *node_ids, double *states);
int mplx_poly_set_record(mplx_poly *p, uint32_t cap_per_query);
int mplx_poly_result_expanded(mplx_poly *p, int32_t q, uint32_t cap, int32_t *ids, uint32_t *n);
int mplx_poly_last_kernel_ms(const mplx_poly *p, float *ms);
/* Look-ahead helper workgroups of mplx_poly_plan_batch (no reference counterpart: the reference runs one planner per
 * thread of control): when every leader workgroup has one query -- the batched tick -- workgroups on the otherwise idle
 * compute units run the collision tests of the states a search has just created, before the search pops them (the
 * outcome is a pure function of the state: results are identical with or without).  per_leader: -1 auto, 0 off, <= 15. */
int mplx_poly_set_helpers(mplx_poly *p, int32_t per_leader);
/* ---- LPA* on the moving-obstacle planner (round 6): PlannerBase::plan with setLPAstar(true), PolyMapPlanner::updateNodes
 *      (mpl_external_planner/.../poly_map_planner/poly_map_planner.h:61-93) and getSubStateSpace, as
 *      mpl_test_node/src/poly_map_replanner_node.cpp:123-186,231 drives them.  One handle = one planner's device-resident state space
 *      (states with g / rhs, predecessor entries with a blocked bit -- EVERY successor get_succ emits has one, the blocked ones too,
 *      as the reference's node records do); worlds, lattice and limits are those of the mplx_poly handle it was created on
 *      (re-commit the world after the obstacles / the start time changed, then call update_nodes).  Time-keyed ACC / JRK states. ---- */
typedef struct mplx_plpa mplx_plpa;
int mplx_plpa_create(mplx_poly *p, mplx_plpa **out);
void mplx_plpa_destroy(mplx_plpa *l);
const char *mplx_plpa_last_error(const mplx_plpa *l);
int mplx_plpa_set_capacity(mplx_plpa *l, uint64_t max_nodes, uint64_t max_edges, uint64_t max_open_log);
int mplx_plpa_initialized(const mplx_plpa *l);   /* PlannerBase::initialized() */
int mplx_plpa_reset(mplx_plpa *l);               /* drop the state space */
/* plan(start, goal) in world `world`; start / goal: pos2 vel2 acc2 jrk2 t.  Repairs and re-uses the previous plan's state space when the
 * goal and the start (= the current root) are unchanged; otherwise starts one. */
int mplx_plpa_plan(mplx_plpa *l, int32_t world, const double *start, const double *goal, double eps, double tol_pos, double tol_vel, int32_t max_expand,
                   int32_t heur_ignore_dynamics, mplx_result *out);
/* updateNodes(): every predecessor entry re-tested against the world as committed now (forward_action + isFree(pr, pred.t)); entries
 * whose outcome changed flip (increaseCost / decreaseCost) and the look-ahead values of their states are recomputed.  Counts: entries
 * that became blocked / free; mplx_plpa_changed lists them by entry number (getBlockedPrimitives / getClearedPrimitives: the entry's
 * parent state and action are in mplx_plpa_result_entries / _nodes). */
int mplx_plpa_update_nodes(mplx_plpa *l, int32_t world, uint64_t *n_blocked, uint64_t *n_cleared);
int mplx_plpa_changed(mplx_plpa *l, uint64_t cap, int32_t *entry, int32_t *now_blocked, uint64_t *n);
/* getSubStateSpace(time_step): re-root at the time_step-th state of the last trajectory, by planning afresh from it (the next plan()
 * from that state finds a consistent space) */
int mplx_plpa_sub_state_space(mplx_plpa *l, int32_t world, int32_t time_step);
int mplx_plpa_traj_len(const mplx_plpa *l);
int mplx_plpa_result_traj(mplx_plpa *l, int32_t *actions, int32_t *node_ids, double *states);
int mplx_plpa_last_kernel_ms(const mplx_plpa *l, float *ms);
/* shader cycles of the last plan's launch by section of an iteration (thread 0's clock): 0 pop, 1 stop test + settling the expanded
 * state, 2 primitives / keys / look-ups / heuristics, 3 isFree of the primitives, 4 link, 5 updateNode of the children (look-ahead values, flags, pushes), 6 goal test + barrier */
int mplx_plpa_result_cycles(const mplx_plpa *l, uint64_t cyc[10]);
int mplx_plpa_counts(const mplx_plpa *l, uint64_t *n_nodes, uint64_t *n_entries);
int mplx_plpa_result_expanded(mplx_plpa *l, uint32_t cap, int32_t *ids, uint32_t *n);
/* state-space dumps (parity tests): per state pos2 vel2 acc2 jrk2 t, g, rhs, h, closed / opened / built flags; per entry, in creation
 * order: child, parent, action, blocked */
int mplx_plpa_result_nodes(mplx_plpa *l, uint64_t cap, double *states, double *g, double *rhs, double *h, int32_t *closed, int32_t *opened, int32_t *built);
int mplx_plpa_result_entries(mplx_plpa *l, uint64_t cap, int32_t *child, int32_t *parent, int32_t *action, int32_t *blocked);
int mplx_poly_last_helpers(const mplx_poly *p); /* helpers per leader of the last launch */
/* Launch guard of the moving-obstacle search: see mplx_set_deadline (a tick that outlives it returns MPLX_ERR_TIMEOUT). */
int mplx_poly_set_deadline(mplx_poly *p, double seconds);
/* shader-clock cycles query q of the last batch spent in [0] pop, [1] get_succ (primitives + collide), [2] look-up + commit */
int mplx_poly_result_cycles(mplx_poly *p, int32_t q, uint64_t cyc[10]);

/* ---- measurement ---- */
/* device-clock begin / end (seconds since the first query of the batch started) and workgroup of query q */
int mplx_result_timing(mplx_ctx *ctx, int q, double *t_begin_s, double *t_end_s, int32_t *slot);
/* shader-clock cycles query q spent in: [0] pop (incl. refill), [1] expand (primitives + voxels),
 * [2] successor look-up (+ commit in the one-node kernel), [3] near-set eviction, [4] refill,
 * [5] coarse-bucket activation, [6] ordered commit; counts: [7] batches, [8] batches committed unit by unit,
 * [9] candidates served from the look-ahead cache.  [0]..[6] are zero in the product build of the library (the timers
 * sit on a query's serial chain; built with -DMPLX_PHASE_TIMERS=1 -- tools/build_variant.sh timers -- they are filled) */
int mplx_result_cycles(mplx_ctx *ctx, int q, uint64_t cyc[10]);
/* Speculation accounting of query q (the K-way speculative kernels; zeros from the one-node kernels): [0] OPEN entries taken as
 * candidates, [1] of those found stale and dropped (a pop the reference's loop would skip too), [2] units that ran get_succ,
 * [3] units whose expansion was thrown away because their batch was cut ahead of them (they return to OPEN and are expanded again).
 * mplx_result.n_expanded counts committed units only: spec[2] - n_expanded is the work -- and traffic -- speculation wasted. */
int mplx_result_speculation(mplx_ctx *ctx, int q, uint64_t spec[4]);
/* duration (ms, HIP events on the context's stream) of the last search / expand kernel launch */
int mplx_last_kernel_ms(const mplx_ctx *ctx, float *ms);
/* name of the search kernel mplx_plan / mplx_plan_batch launches for the current configuration */
const char *mplx_kernel_name(const mplx_ctx *ctx);
/* Counter bumped by every mplx_plan / mplx_plan_batch on this context.  The result getters answer for the LAST
 * plan of the context; host wrappers that share one context between several planner objects (the reference
 * shares one MapUtil between two planners, map_replanner_node.cpp:415,427) remember the epoch of their own
 * plan and refuse to answer from another planner's state space. */
uint64_t mplx_plan_epoch(const mplx_ctx *ctx);
const char *mplx_version(void);

/* ---- after the search: refinement and sampling (host arithmetic, no context, no device) ----
 * TrajSolver3D(control).setWaypoints(wps).setDts(dts).solve(), map_planner_node.cpp:217-227: minimum-derivative
 * piecewise polynomial through n_wp waypoints (control kind VEL / ACC / JRK: minimum velocity / acceleration / jerk);
 * the control bits of every waypoint say which of its derivatives are fixed (the node sets the intermediate ones to
 * VEL = position only).  prs: n_wp - 1 primitives out.  MPLX_ERR_ARG: fewer than 2 waypoints, a time <= 0, a singular
 * system, or MPLX_SNP (septic segments do not fit a Primitive: traj_solver_node.cpp:67 "does not work"). */
int mplx_traj_solve(int32_t control, int32_t n_wp, const mplx_waypoint *wps, const double *dts, mplx_primitive *prs);
/* Trajectory::sample(N), trajectory_extractor.hpp:9-30: N + 1 equally spaced states of the piecewise trajectory
 * (pos / vel / acc / jrk / yaw / t filled), yaw_dot (N + 1, may be NULL) */
int mplx_traj_sample(int32_t n_prs, const mplx_primitive *prs, int32_t N, mplx_waypoint *out, double *yaw_dot);
/* Trajectory::J(control) summed over the primitives (map_planner_node.cpp:210-214); control MPLX_YAW: Jyaw */
double mplx_traj_effort(int32_t n_prs, const mplx_primitive *prs, int32_t control);

#ifdef __cplusplus
}
#endif
#endif
