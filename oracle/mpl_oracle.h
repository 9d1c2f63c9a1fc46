/*
 * mpl_oracle.h -- CPU restatement of the motion_primitive_library v1.2 voxel-map search path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and there only as the
 * checker / the reported CPU baseline.  The product path (mpl_ros_amd/csrc) shares no code with
 * this directory.
 *
 * PARITY UNPINNED.  The reference tree (/root/reference) does not contain the algorithm: it lives
 * in the un-vendored git submodule `motion_primitive_library` ("v1.2", reference README.md:4,
 * .gitmodules:1-3; exact commit unrecoverable).  The reference also holds no tests, golden vectors
 * or known answers for this path.  This file therefore restates the
 * published algorithm, anchored on what IS citable in-tree:
 *   - polynomial convention p(t)=c0/120 t^5+c1/24 t^4+c2/6 t^3+c3/2 t^2+c4 t+c5
 *       mpl_external_planner/include/mpl_external_planner/poly_map_planner/primitive_geometry_utils.h:12-26,135-146
 *   - get_succ control flow (clear; push expanded; for i in U: build, evaluate, filter, cost)
 *       .../poly_map_planner/env_poly_map.h:45-69 and .../ellipsoid_planner/env_cloud.h:50-70
 *   - sampling density n = ceil(max_v * t / res)            .../ellipsoid_planner/ellipsoid_util.h:67-70
 *   - grid layout idx = x + dx*y + dx*dy*z, free 0 / occ 100 / unknown -1
 *       planning_ros_utils/src/mapping_utils/voxel_grid.cpp:88, include/planning_ros_utils/voxel_grid.h:43-45
 *   - node record pred_coord / pred_action_id / pred_action_cost   .../poly_map_planner/poly_map_planner.h:70-86
 *   - control-set generation and start/goal construction      mpl_test_node/src/map_planner_node.cpp:108-171
 * Everything else (hash resolutions, floatToInt rounding, validate_primitive, heuristic, A* loop
 * order, heap comparator, defaults) is a recollection of upstream MPL and is tagged UNVERIFIED in
 * mpl_oracle.c.  It is pinned only by analytic known answers (tests/test_oracle_kat.py).
 *
 * The VoxelGrid part (orc_grid_*) and MapUtil::getCloud's loop order are different: their reference
 * source IS in-tree (planning_ros_utils/src/mapping_utils/voxel_grid.cpp) and is restated line by line,
 * tagged IN-TREE with line ranges.  Its own build needs catkin, Eigen, boost::multi_array and generated
 * message headers, all absent; `make -C oracle ref` compiles the reference's voxel_grid.cpp from where it lies
 * against stand-ins for those three headers (oracle/ref_stubs/ + include/mpl_shim typedefs) into
 * oracle/_ref/libvoxelgrid_ref.so, and tests/test_voxel_grid.py checks the restatement -- and, on the GPU,
 * the HIP grid -- against that binary bit for bit.  For this component parity is pinned by the reference.
 *
 * DEVIATIONS FROM THE RECOLLECTED UPSTREAM -- the complete list, so that it can be diffed against
 * motion_primitive_library the day its source is available.  Each is tagged [DEVIATION Dn] (or plain
 * [DEVIATION]) at the line it applies to in mpl_oracle.c; the HIP product mirrors every one of them.
 *   D1  polynomial roots (heuristic): upstream Cardano/Ferrari (acos, cos, cbrt) for degree 3-4 and Eigen
 *       companion-matrix eigenvalues for 5-6; here derivative-chain isolation + safeguarded Newton/bisection
 *       from + - * / sqrt only (bit-reproducible on host and device).  Same real roots to ~1e-15 relative.
 *   D2  Primitive::J: upstream expands the integral into a fixed closed form; here the double sum over the
 *       derivative's monomial coefficients in ascending (i, j) order.  Equal up to f64 rounding.
 *   D3  node identity: the quantised integer key tuple itself, not boost::hash_combine of it (upstream
 *       operator== compares hash values, so two states whose 64-bit hashes collide would merge upstream).
 *   D4  is_free(pr) with n == 0 samples (stationary primitive): upstream would evaluate dt = t/0; here the
 *       single sample t = 0 is tested.
 *   D5  OPEN tie-breaking: upstream's d_ary_heap compares (f, then g); remaining ties are heap-internal and
 *       unspecified.  Here the strict total order (f, g, node id), id = order of first finite arrival.
 *   D6  closed node improved (inconsistent heuristic, eps > 1 only): re-opened.  Upstream prints
 *       "ASTAR ERROR!"; whether it re-pushes is UNVERIFIED.  Never happens at eps = 1 (n_reopen = 0 in all
 *       BASELINE configs).
 *   D7  successors with cost +inf (blocked primitives): upstream gives them an hm_ entry and an inf-cost pred
 *       entry; here they live in a side list (parent, action) and orc_num_states_all / orc_get_blocked_edges
 *       rebuild hm_.size() / the inf-cost pred entries on request.  Expansion order, g values, trajectory and
 *       cost are unaffected (an inf-cost edge never relaxes anything); node ids are "order of first finite
 *       arrival".
 *   D8  env_base::cal_heur with v_max <= 0 ("unlimited"): the |dp|_inf / v_max arrival-time bound is dropped
 *       (t_bar = 0; heur_ignore_dynamics returns w |dp|_inf) instead of dividing by a non-positive number.
 *       Believed to equal upstream's `v_max_ > 0` guard; UNVERIFIED.
 *
 * OPEN QUESTIONS -- places where a reviewer's recollection of upstream differs from this restatement and neither can be
 * checked against source (the submodule is empty); listed so that they are the first things diffed when it is available.
 * They are NOT known deviations: each is tagged [UNVERIFIED Qn] at the line it applies to; the HIP product follows this file.
 *   Q1  Primitive1D::p / v / a / j(t): upstream is recalled to raise t to its powers with std::pow(t, n); here --
 *       and on the device -- the powers are repeated multiplications (t * t * t ...).  Identical at t = dt = 1 (every end
 *       state of BASELINE C1-C4) and wherever pow's result is the correctly rounded product chain; at the interior sample
 *       times i dt / n of a JRK primitive (cubic / quartic terms) the two can differ in the last ulp of a position, which
 *       matters only if the position sits within an ulp of a voxel boundary.  No cell index in any committed fixture
 *       depends on it as far as the analytic pins (tests/test_oracle_kat.py) can tell; unpinned either way.
 *   Q2  potential mask (P1 of mpl_oracle_pot.inc): upstream is recalled to normalise the distance by the INTEGER radius
 *       rn = ceil(r / res) in cells (d = |n| / rn per axis), this file normalises by the metric radius (n res / r).  The
 *       two agree when r is a multiple of res (1.5 / 0.1 in distance_map_planner_node.cpp:187 -- up to the rounding of
 *       1.5 / 0.1); they differ for other radii.
 *   Q3  traverse cost of the potential (P3): upstream is recalled to walk t in [0, T) in steps of dt_sample and to scale
 *       the summed potential by that step (an approximate line integral), this file sums the potential over the n + 1
 *       collision samples unscaled.  The weight the reference passes (10) then means different things in the two forms;
 *       path shape under a potential field is therefore NOT claimed to match upstream, only the restatement.
 */
#ifndef MPL_ORACLE_H
#define MPL_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Control bit flags (UNVERIFIED recollection of mpl_basis/control.h): a Waypoint's control is the
 * union of its use_pos/use_vel/use_acc/use_jrk/use_yaw bits. */
enum { ORC_VEL = 1, ORC_ACC = 3, ORC_JRK = 7, ORC_SNP = 15, ORC_YAW = 16 /* use_yaw bit: Control::*xYAW = base | 16 */ };

typedef struct {
  double pos[3], vel[3], acc[3], jrk[3];
  double yaw, t;
  int32_t control;  /* ORC_VEL / ORC_ACC / ORC_JRK / ORC_SNP, | ORC_YAW when the state carries yaw (use_yaw) */
  int32_t enable_t; /* time is part of the key (false for env_map) */
} orc_waypoint;

typedef struct {
  double c[3][6]; /* per-axis coefficients, convention above */
  double cyaw[6]; /* yaw channel (Primitive::pr_yaw): a VEL-type primitive, yaw(t) = cyaw[4] t + cyaw[5] */
  double t;
  int32_t control;
  int32_t pad;
} orc_primitive;

typedef struct {
  int32_t control;   /* control kind of the search states */
  int32_t n_u;       /* number of control inputs */
  const double *U;   /* n_u x 3 */
  double dt, v_max, a_max, j_max;
  double w;          /* time weight, default 10 */
  double eps;        /* heuristic weight, default 1 */
  double tol_pos, tol_vel, tol_acc; /* <0 disables vel/acc */
  double t_max;      /* +inf unless set */
  int32_t max_expand; /* <=0: unlimited */
  int32_t heur_ignore_dynamics;
  const double *U_yaw; /* n_u yaw rates: the 4th component of the Vec4f control inputs (map_planner_node.cpp:119-139); NULL: none */
  double yaw_max;      /* setYawmax; <= 0 disables validate_yaw */
  double tol_yaw;      /* < 0 disables the yaw term of is_goal */
} orc_config;

typedef struct {
  uint64_t n_expansions;  /* get_succ calls */
  uint64_t n_voxel_reads; /* map_[idx] loads inside is_free(pr), early-out honoured */
  uint64_t n_primitives;  /* primitives built */
  uint64_t n_succ;        /* successors emitted (finite or inf cost) */
  uint64_t n_succ_finite; /* successors with finite cost */
  uint64_t n_new_nodes;   /* nodes created in the state space */
  uint64_t n_heap_push, n_heap_decrease, n_reopen;
} orc_counters;

typedef struct orc_planner orc_planner;

/* ---- basis (a3,a4,a5,a6,a9 of SURVEY 8a) ---- */
void orc_primitive_build(const orc_waypoint *p, const double *u, double dt, orc_primitive *out);
/* with the yaw rate of a Vec4f control input (the state carries yaw: control & ORC_YAW) */
void orc_primitive_build_yaw(const orc_waypoint *p, const double *u, double u_yaw, double dt, orc_primitive *out);
int orc_validate_yaw(const orc_primitive *pr, double yaw_max);
void orc_det_sincos(double x, double *sn, double *cs); /* the fixed + - * / sequence validate_yaw evaluates sin / cos with */
void orc_primitive_evaluate(const orc_primitive *pr, double t, orc_waypoint *out);
double orc_primitive_max_vel(const orc_primitive *pr, int k);
double orc_primitive_max_acc(const orc_primitive *pr, int k);
double orc_primitive_max_jrk(const orc_primitive *pr, int k);
double orc_primitive_J(const orc_primitive *pr, int control);
int orc_validate_primitive(const orc_primitive *pr, double mv, double ma, double mj);
/* quantised key of a waypoint: writes up to 13 ints, returns how many */
int orc_waypoint_key(const orc_waypoint *w, int32_t *key);
/* real roots in (lo, +inf) of sum a[i] x^i, i=0..n, ascending; returns count (<= n) */
int orc_poly_roots_above(const double *a, int n, double lo, double *roots);

/* ---- planner ---- */
orc_planner *orc_create(void);
void orc_destroy(orc_planner *);
/* copy-in map; values: free 0, occupied >0, unknown -1 */
void orc_set_map(orc_planner *, const int8_t *data, const int32_t dim[3], const double origin[3], double res);
/* adopt the caller's read-only grid without a copy (one map shared by many planner objects / threads) */
void orc_set_map_shared(orc_planner *, const int8_t *data, const int32_t dim[3], const double origin[3], double res);
void orc_free_unknown(orc_planner *);
/* MapUtil helpers next to the search */
void orc_map_dilate(orc_planner *, int n_offsets, const int32_t *offsets);
void orc_map_get(const orc_planner *, int8_t *out);
int orc_map_cell_state(const orc_planner *, const int32_t pn[3]);
int orc_map_raytrace(const orc_planner *, const double p1[3], const double p2[3], int32_t *cells, int cap);
uint64_t orc_map_cloud(const orc_planner *, int which, double *pts, uint64_t cap);
void orc_set_config(orc_planner *, const orc_config *cfg);
void orc_set_goal(orc_planner *, const orc_waypoint *goal);
void orc_float_to_int(const orc_planner *, const double pt[3], int32_t pn[3]);
int orc_is_free_point(const orc_planner *, const double pt[3]);
int orc_is_free_primitive(orc_planner *, const orc_primitive *pr);
double orc_heuristic(const orc_planner *, const orc_waypoint *state);
int orc_is_goal(const orc_planner *, const orc_waypoint *state);
/* env_map::get_succ; arrays sized n_u; returns number of successors emitted */
int orc_get_succ(orc_planner *, const orc_waypoint *curr, orc_waypoint *succ, double *succ_cost, int32_t *action_idx);

/* status codes shared with the product C-ABI */
enum { ORC_OK = 0, ORC_NO_PATH = 1, ORC_START_OCCUPIED = 2, ORC_MAX_EXPAND = 3 };
int orc_plan(orc_planner *, const orc_waypoint *start, const orc_waypoint *goal);
double orc_traj_cost(const orc_planner *);
/* results of the last plan(): sizes then copy-out */
int orc_num_expanded(const orc_planner *);                 /* expansion sequence length */
void orc_get_expanded(const orc_planner *, int32_t *node_ids, double *pos /* n x 3 */);
uint64_t orc_expand_hash(const orc_planner *);               /* fold of the expansion sequence (node ids) */
int orc_num_nodes(const orc_planner *);
void orc_get_node(const orc_planner *, int id, orc_waypoint *coord, double *g, double *h, int32_t *closed);
int orc_num_closed(const orc_planner *);
/* predecessor lists: for every node in id order its edges in arrival order; returns the number of edges */
int orc_get_edges(const orc_planner *, int32_t *child, int32_t *parent, int32_t *action, int cap);
/* successors emitted with +inf cost (deviation D7): count, (parent id, action) in arrival order, and the
 * size upstream's hm_ would have (finite states + states only reached by blocked primitives) */
int orc_num_blocked(const orc_planner *);
void orc_get_blocked_edges(const orc_planner *, int32_t *parent, int32_t *action);
int orc_num_states_all(const orc_planner *);
int orc_traj_len(const orc_planner *);                      /* number of primitives */
void orc_get_traj(const orc_planner *, orc_primitive *prs, orc_waypoint *wps /* len+1 */, int32_t *actions, int32_t *node_ids /* len+1 */);
void orc_get_counters(const orc_planner *, orc_counters *);
void orc_reset_counters(orc_planner *);

/* ---- potential-field cost and search region of the map planner (mpl_oracle_pot.inc; UNVERIFIED restatement anchored on
 *      distance_map_planner_node.cpp:185-193,218-224,231) ---- */
void orc_set_potential_weights(orc_planner *, double potential_weight, double gradient_weight);
void orc_potential_update(orc_planner *, const double radius[3], const double pos[3], const double range[3], int pow_);
void orc_search_region_set(orc_planner *, int n_pts, const double *pts, const double radius[3], int dense);
void orc_potential_clear(orc_planner *);
void orc_aux_get(const orc_planner *, int8_t *out); /* 0..100 potential, -1 outside the search region */

/* ---- LPA* incremental replanning (mpl_oracle_lpa.inc; UNVERIFIED restatement anchored on map_replanner_node.cpp:107-255,
 *      425-437 and poly_map_planner.h:61-93).  With orc_set_lpastar(p, 1) orc_plan() keeps and repairs its state space. ---- */
void orc_set_lpastar(orc_planner *, int on);                /* PlannerBase::setLPAstar */
int orc_lpa_initialized(const orc_planner *);               /* PlannerBase::initialized() */
int orc_lpa_update_blocked(orc_planner *, int n_cells, const int32_t *cells); /* MapPlanner::updateBlockedNodes; entries changed */
int orc_lpa_update_cleared(orc_planner *, int n_cells, const int32_t *cells); /* MapPlanner::updateClearedNodes */
void orc_lpa_sub_state_space(orc_planner *, int time_step); /* PlannerBase::getSubStateSpace */
void orc_lpa_set_reroot(orc_planner *, int mode);              /* how: 0 Dijkstra through the expanded states (L5), 1 plan afresh from the k-th state (L5b), 2 auto (default) */
int orc_lpa_iterations(const orc_planner *);                /* states popped by the last plan() */
double orc_get_node_rhs(const orc_planner *, int id);
int orc_get_node_opened(const orc_planner *, int id);
int orc_get_edges_blocked(const orc_planner *, int32_t *blocked, int cap);

/* ---- VoxelGrid (in-tree planning_ros_utils/src/mapping_utils/voxel_grid.cpp, restated line by line) ---- */
typedef struct orc_grid orc_grid;
orc_grid *orc_grid_create(const double origin[3], const double dim[3], float res);
void orc_grid_destroy(orc_grid *);
int orc_grid_allocate(orc_grid *, const double new_dim_d[3], const double new_ori_d[3]);
void orc_grid_info(const orc_grid *, int32_t dim[3], double origin_d[3], float *res);
void orc_grid_clear(orc_grid *);
void orc_grid_add_cloud(orc_grid *, int n, const double *pts);
int orc_grid_add_cloud_ns(orc_grid *, int n, const double *pts, int n_ns, const int32_t *ns, int32_t *new_obs, int cap);
void orc_grid_decay(orc_grid *);
void orc_grid_clear_column(orc_grid *, int nx, int ny);
void orc_grid_fill_column(orc_grid *, int nx, int ny);
void orc_grid_fill_cell(orc_grid *, int nx, int ny, int nz);
void orc_grid_get_map(const orc_grid *, int inflated, int8_t *data);
uint64_t orc_grid_get_cloud(const orc_grid *, double *pts, uint64_t cap);

/* Audits of the open questions Q1 / Q2 above (tests/test_oracle_kat.py): how much of a given search / mask depends on them.
 * orc_q1_audit(1) starts counting (single-threaded use), orc_q1_counts: {collision samples evaluated both ways, samples whose
 * position differs in any bit, samples whose CELL differs}.  orc_q2_mask_audit: {entries of this file's potential mask, entries
 * of the integer-radius form, offsets whose value differs}. */
void orc_q1_audit(int on);
void orc_q1_counts(uint64_t out[3]);
void orc_q2_mask_audit(double res, const double radius[3], int pw, uint64_t out[3]);

#ifdef __cplusplus
}
#endif
#endif
