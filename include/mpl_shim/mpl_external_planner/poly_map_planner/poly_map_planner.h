/**
 * @file poly_map_planner.h  (mplx shim of <mpl_external_planner/poly_map_planner/poly_map_planner.h>)
 *
 * MPL::PolyMapPlanner<Dim> with the search done on the GPU (libmplx.so: mplx_poly_*).  Put include/mpl_shim AHEAD of the
 * reference's mpl_external_planner/include on the include path: this file then replaces the reference's 53-line planner
 * glue of the same name, and everything else of that directory -- env_poly_map.h, poly_map_util.h,
 * primitive_geometry_utils.h, simple_obstacle.h -- is still the reference's own, unchanged (it is included below from
 * where it lies).  Class name, method names and argument meaning are the reference's (poly_map_planner.h:17-93), so its
 * callers -- mpl_test_node/src/robot.hpp:92-133 (planner_ptr.reset(new MPL::PolyMapPlanner<Dim>(false)) ... plan(start_,
 * goal_)), robot_team.hpp, poly_map_planner_node.cpp -- compile and run unchanged.
 *
 * Why a replacement and not a hook: the reference's planner searches through ENV_, a host C++ object whose get_succ is a
 * virtual call per expansion; a device search cannot call it, and PolyMapUtil keeps its obstacles private.  Here the
 * setters keep a copy of the obstacles next to handing them to the reference's PolyMapUtil (which still answers
 * getPolyhedrons / getBoundingBox / getLinearObstacles and is what ENV_ wraps), and plan() uploads them as ONE world of
 * an mplx_poly object and runs PlannerBase::plan there (time-keyed states, env_poly_map.h:63-64; cost
 * J + 0.001 J(VEL) + w dt, :71-73).  2-D only, like every in-tree caller (multi_robot_node.cpp, poly_map_planner_node.cpp).
 * With setLPAstar(true) (poly_map_replanner_node.cpp:352) the planner keeps a device-resident LPA* state space of its own
 * (mplx_plpa_*): updateNodes() (:61-93) re-tests every stored predecessor primitive against the obstacles as they are now and
 * fills getBlockedPrimitives / getClearedPrimitives, plan() repairs, getSubStateSpace(k) re-roots (round 6).
 */
#ifndef MPLX_SHIM_POLY_MAP_PLANNER_H
#define MPLX_SHIM_POLY_MAP_PLANNER_H

#include <mpl_external_planner/poly_map_planner/env_poly_map.h>  // the reference's own (next on the include path)
#include <mpl_planner/common/planner_base.h>
#include <mplx.h>

#include <vector>

namespace MPL {

/// One device object serves every planner of the process: the reference's robots construct a NEW PolyMapPlanner for
/// every plan() (robot.hpp:109), and device pools are not something to allocate at 2 Hz per robot.
inline mplx_poly *shared_poly_device(bool destroy = false) {
  static mplx_poly *p = nullptr;
  if (destroy) { if (p) mplx_poly_destroy(p); p = nullptr; return nullptr; }
  if (!p) {
    if (mplx_poly_create(0, &p) != MPLX_OK) {
      printf(ANSI_COLOR_RED "[PolyMapPlanner] %s\n" ANSI_COLOR_RESET, mplx_poly_last_error(nullptr));
      p = nullptr;
    }
  }
  return p;
}
/// pool capacities of that object (states, predecessor records, OPEN-log entries of one search); doubled on MPLX_PLAN_POOL_FULL
inline uint64_t *shared_poly_capacity() {
  static uint64_t cap[3] = {1u << 20, 1u << 22, 1u << 21};
  return cap;
}

template <int Dim>
class PolyMapPlanner : public PlannerBase<Dim, Waypoint<Dim>> {
  typedef PlannerBase<Dim, Waypoint<Dim>> Base;

 public:
  PolyMapPlanner(bool verbose = false) {
    this->planner_verbose_ = verbose;
    if (this->planner_verbose_) printf(ANSI_COLOR_CYAN "[PolyMapPlanner] PLANNER VERBOSE ON (mplx back-end)\n" ANSI_COLOR_RESET);
  }
  /// Set map util (poly_map_planner.h:30-34)
  void setMap(const Vecf<Dim> &ori, const Vecf<Dim> &dim) {
    map_util_.reset(new PolyMapUtil<Dim>());
    map_util_->setBoundingBox(ori, dim);
    this->ENV_.reset(new MPL::env_poly_map<Dim>(map_util_));
    this->apply_to_env();  // (the reference's setters forward to ENV_, which exists only from here on)
    ori_ = ori;
    dim_ = dim;
    has_map_ = true;
  }
  void setStartTime(decimal_t t) { map_util_->setStartTime(t); start_t_ = t; }
  void setStaticObstacles(const vec_E<PolyhedronObstacle<Dim>> &obs) { map_util_->setStaticObstacle(obs); static_obs_ = obs; }
  void setLinearObstacles(const vec_E<PolyhedronLinearObstacle<Dim>> &obs) { map_util_->setLinearObstacle(obs); linear_obs_ = obs; }
  void setNonlinearObstacles(const vec_E<PolyhedronNonlinearObstacle<Dim>> &obs) { map_util_->setNonlinearObstacle(obs); nonlinear_obs_ = obs; }
  vec_E<Polyhedron<Dim>> getPolyhedrons(decimal_t time) const { return map_util_->getPolyhedrons(time); }
  vec_E<PolyhedronLinearObstacle<Dim>> getLinearObstacles() const { return map_util_->getLinearObstacles(); }
  Polyhedron<Dim> getBoundingBox() const { return map_util_->getBoundingBox(); }

  ~PolyMapPlanner() {
    if (lpa_) mplx_plpa_destroy(lpa_);
  }
  /// poly_map_planner.h:61-93: after setLinearObstacles / setStartTime, re-test every stored predecessor primitive of the kept LPA*
  /// state space (forward_action + isFree(pr, pred.t)); what became blocked / free is collected, costs change, look-ahead values follow
  void updateNodes() {
    blocked_prs_.clear();
    cleared_prs_.clear();
    if (!lpa_ || !mplx_plpa_initialized(lpa_)) return;  // (if (!this->ss_ptr_) return;)
    mplx_poly *p = shared_poly_device();
    if (!p || !upload(p, lpa_control_)) return;
    uint64_t nb = 0, nc = 0, n = 0;
    if (!lcheck(mplx_plpa_update_nodes(lpa_, 0, &nb, &nc))) return;
    if (!lcheck(mplx_plpa_changed(lpa_, 0, nullptr, nullptr, &n)) || n == 0) return;
    std::vector<int32_t> entry((size_t)n), now((size_t)n);
    uint64_t nn = 0, ne = 0;
    if (!lcheck(mplx_plpa_changed(lpa_, n, entry.data(), now.data(), &n)) || !lcheck(mplx_plpa_counts(lpa_, &nn, &ne))) return;
    std::vector<double> states((size_t)nn * 9);
    std::vector<int32_t> parent((size_t)ne), action((size_t)ne);
    if (!lcheck(mplx_plpa_result_nodes(lpa_, nn, states.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr))) return;
    if (!lcheck(mplx_plpa_result_entries(lpa_, ne, nullptr, parent.data(), action.data(), nullptr))) return;
    for (size_t i = 0; i < (size_t)n; i++) {  // the primitive of a changed entry: forward_action(pred_coord, pred_action_id)
      const double *st = &states[(size_t)parent[(size_t)entry[i]] * 9];
      Waypoint<Dim> w((Control::Control)lpa_control_);
      for (int k = 0; k < 2; k++) { w.pos(k) = st[k]; w.vel(k) = st[2 + k]; w.acc(k) = st[4 + k]; w.jrk(k) = st[6 + k]; }
      w.t = st[8];
      const Primitive<Dim> pr(w, this->U_vec_[(size_t)action[(size_t)entry[i]]], this->dt_);
      (now[i] ? blocked_prs_ : cleared_prs_).push_back(pr);
    }
  }
  /// PlannerBase::initialized() / getSubStateSpace(time_step) of the LPA* planner (poly_map_replanner_node.cpp:231)
  bool initialized() { return lpa_ && mplx_plpa_initialized(lpa_); }
  void getSubStateSpace(int time_step) {
    if (!lpa_ || !mplx_plpa_initialized(lpa_)) return;
    mplx_poly *p = shared_poly_device();
    if (!p || !upload(p, lpa_control_)) return;
    lcheck(mplx_plpa_sub_state_space(lpa_, 0, time_step));
  }
  vec_E<Primitive<Dim>> getBlockedPrimitives() { return blocked_prs_; }
  vec_E<Primitive<Dim>> getClearedPrimitives() { return cleared_prs_; }

  /// bool PlannerBase::plan(start, goal) (robot.hpp:123, poly_map_planner_node.cpp): on the device, through mplx_poly_*
  bool plan(const Waypoint<Dim> &start, const Waypoint<Dim> &goal) override {
    if (this->planner_verbose_) { start.print("Start:"); goal.print("Goal:"); }
    this->traj_ = Trajectory<Dim>();
    this->traj_cost_ = std::numeric_limits<decimal_t>::infinity();
    res_ = mplx_result();
    if (Dim != 2 || !has_map_) {
      printf(ANSI_COLOR_RED "[PolyMapPlanner] plan() refused: the mplx back-end plans 2-D moving-obstacle searches after setMap()\n" ANSI_COLOR_RESET);
      return false;
    }
    mplx_poly *p = shared_poly_device();
    if (!p) return false;
    const int32_t control = (int32_t)start.control & 15;
    if (!upload(p, control)) return false;
    double s9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, g9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 2; i++) {
      s9[i] = start.pos(i); s9[2 + i] = start.vel(i); s9[4 + i] = start.acc(i); s9[6 + i] = start.jrk(i);
      g9[i] = goal.pos(i); g9[2 + i] = goal.vel(i); g9[4 + i] = goal.acc(i); g9[6 + i] = goal.jrk(i);
    }
    s9[8] = start.t;
    const int32_t world = 0;
    uint64_t *cap = shared_poly_capacity();
    if (this->use_lpastar_) return plan_lpastar(start, s9, g9, control);
    for (int attempt = 0;; attempt++) {
      if (!check(p, mplx_poly_set_capacity(p, 1, cap[0], cap[1], cap[2]))) return false;
      if (!check(p, mplx_poly_plan_batch(p, 1, &world, s9, g9, this->epsilon_, this->tol_pos_, this->tol_vel_, this->max_num_,
                                         this->heur_ignore_dynamics_ ? 1 : 0, &res_))) return false;
      if (res_.status != MPLX_PLAN_POOL_FULL || attempt >= 6) break;
      for (int k = 0; k < 3; k++) cap[k] *= 2;  // (the reference grows std containers: grow the device pools and search again)
      printf(ANSI_COLOR_CYAN "[PolyMapPlanner] device pools exhausted: doubled, planning again\n" ANSI_COLOR_RESET);
    }
    if (res_.status == MPLX_PLAN_START_OCCUPIED) { printf(ANSI_COLOR_RED "[PlannerBase] start is not free!\n" ANSI_COLOR_RESET); return false; }
    this->traj_cost_ = res_.cost;
    if (res_.status != MPLX_PLAN_OK || std::isinf(res_.cost)) {
      printf(ANSI_COLOR_RED "[MPPlanner] Cannot find a traj! (status %d)\n" ANSI_COLOR_RESET, res_.status);
      this->traj_cost_ = std::numeric_limits<decimal_t>::infinity();
      return false;
    }
    const int len = res_.traj_len;
    std::vector<int32_t> actions((size_t)(len > 0 ? len : 1)), ids((size_t)len + 1);
    std::vector<double> states((size_t)(len + 1) * 9);
    if (!check(p, mplx_poly_result_traj(p, 0, actions.data(), ids.data(), states.data()))) return false;
    vec_E<Primitive<Dim>> prs;
    for (int i = 0; i < len; i++) {
      Waypoint<Dim> w(start.control);
      const double *s = &states[(size_t)i * 9];
      for (int k = 0; k < 2; k++) { w.pos(k) = s[k]; w.vel(k) = s[2 + k]; w.acc(k) = s[4 + k]; w.jrk(k) = s[6 + k]; }
      w.t = s[8];
      prs.push_back(Primitive<Dim>(w, this->U_vec_[(size_t)actions[(size_t)i]], this->dt_));
    }
    this->traj_ = Trajectory<Dim>(prs);
    return true;
  }
  const mplx_result &getResult() const { return res_; }
  size_t getExpandedNum() const { return (size_t)res_.n_expanded; }

 protected:
  static std::vector<double> planes(const Polyhedron<Dim> &poly) {  // n_hp x {px, py, nx, ny}
    std::vector<double> hp;
    for (const auto &v : poly.hyperplanes()) { hp.push_back(v.p_(0)); hp.push_back(v.p_(1)); hp.push_back(v.n_(0)); hp.push_back(v.n_(1)); }
    return hp;
  }
  /// the world this planner sees -- bounding box, start time, the three obstacle kinds -- as world 0 of the shared device object
  bool upload(mplx_poly *p, int32_t control) {
    std::vector<double> U;
    for (const auto &u : this->U_vec_) { U.push_back(u(0)); U.push_back(u(1)); }
    const int n_u = (int)this->U_vec_.size();
    if (!check(p, mplx_poly_config(p, control, n_u, U.data(), this->dt_, this->v_max_, this->a_max_, this->j_max_, this->w_))) return false;
    if (!check(p, mplx_poly_begin(p, 1))) return false;
    const double ori[2] = {ori_(0), ori_(1)}, dim[2] = {dim_(0), dim_(1)};
    if (!check(p, mplx_poly_set_world(p, 0, ori, dim, start_t_))) return false;
    for (const auto &o : static_obs_) {
      const std::vector<double> hp = planes(o.geometry());
      const double pt[2] = {o.p()(0), o.p()(1)};
      if (!check(p, mplx_poly_add_static(p, 0, (int32_t)(hp.size() / 4), hp.data(), pt))) return false;
    }
    for (const auto &o : linear_obs_) {
      const std::vector<double> hp = planes(o.geometry());
      const double pt[2] = {o.p()(0), o.p()(1)}, v[2] = {o.v()(0), o.v()(1)};
      if (!check(p, mplx_poly_add_linear(p, 0, (int32_t)(hp.size() / 4), hp.data(), pt, v, o.cov_v()))) return false;
    }
    for (const auto &o : nonlinear_obs_) {
      const std::vector<double> hp = planes(o.geometry());
      std::vector<double> segs;  // n_seg x {cx[6], cy[6], T}
      for (const auto &pr : o.traj().getPrimitives()) {
        for (int ax = 0; ax < 2; ax++) {
          const Vec6f c = pr.pr(ax).coeff();
          for (int k = 0; k < 6; k++) segs.push_back(c(k));
        }
        segs.push_back(pr.t());
      }
      if (!check(p, mplx_poly_add_nonlinear(p, 0, (int32_t)(hp.size() / 4), hp.data(), (int32_t)(segs.size() / 13), segs.data(), o.start_t(),
                                            o.disappear_front_ ? 1 : 0, o.disappear_back_ ? 1 : 0))) return false;
    }
    return check(p, mplx_poly_commit(p));
  }
  /// plan() with setLPAstar(true): on the planner's own device-resident state space (mplx_plpa_*)
  bool plan_lpastar(const Waypoint<Dim> &start, const double *s9, const double *g9, int32_t control) {
    if (!lpa_) {
      if (mplx_plpa_create(shared_poly_device(), &lpa_) != MPLX_OK) { lpa_ = nullptr; return false; }
      const uint64_t *cap = shared_poly_capacity();
      mplx_plpa_set_capacity(lpa_, cap[0] / 4, cap[1] / 4, cap[2] / 2);
    }
    lpa_control_ = control;
    if (!lcheck(mplx_plpa_plan(lpa_, 0, s9, g9, this->epsilon_, this->tol_pos_, this->tol_vel_, this->max_num_, this->heur_ignore_dynamics_ ? 1 : 0, &res_))) return false;
    if (res_.status == MPLX_PLAN_START_OCCUPIED) { printf(ANSI_COLOR_RED "[PlannerBase] start is not free!\n" ANSI_COLOR_RESET); return false; }
    this->traj_cost_ = res_.cost;
    if (res_.status != MPLX_PLAN_OK || std::isinf(res_.cost)) {
      printf(ANSI_COLOR_RED "[MPPlanner] Cannot find a traj! (status %d)\n" ANSI_COLOR_RESET, res_.status);
      this->traj_cost_ = std::numeric_limits<decimal_t>::infinity();
      return false;
    }
    const int len = mplx_plpa_traj_len(lpa_);
    std::vector<int32_t> actions((size_t)(len > 0 ? len : 1)), ids((size_t)len + 1);
    std::vector<double> states((size_t)(len + 1) * 9);
    if (!lcheck(mplx_plpa_result_traj(lpa_, actions.data(), ids.data(), states.data()))) return false;
    vec_E<Primitive<Dim>> prs;
    for (int i = 0; i < len; i++) {
      Waypoint<Dim> w(start.control);
      const double *s = &states[(size_t)i * 9];
      for (int k = 0; k < 2; k++) { w.pos(k) = s[k]; w.vel(k) = s[2 + k]; w.acc(k) = s[4 + k]; w.jrk(k) = s[6 + k]; }
      w.t = s[8];
      prs.push_back(Primitive<Dim>(w, this->U_vec_[(size_t)actions[(size_t)i]], this->dt_));
    }
    this->traj_ = Trajectory<Dim>(prs);
    return true;
  }
  bool lcheck(int rc) {
    if (rc == MPLX_OK) return true;
    printf(ANSI_COLOR_RED "[PolyMapPlanner] %s\n" ANSI_COLOR_RESET, mplx_plpa_last_error(lpa_));
    return false;
  }
  static bool check(mplx_poly *p, int rc) {
    if (rc == MPLX_OK) return true;
    printf(ANSI_COLOR_RED "[PolyMapPlanner] %s\n" ANSI_COLOR_RESET, mplx_poly_last_error(p));
    return false;
  }
  std::shared_ptr<PolyMapUtil<Dim>> map_util_;
  vec_E<Primitive<Dim>> blocked_prs_;
  vec_E<Primitive<Dim>> cleared_prs_;
  vec_E<PolyhedronObstacle<Dim>> static_obs_;
  vec_E<PolyhedronLinearObstacle<Dim>> linear_obs_;
  vec_E<PolyhedronNonlinearObstacle<Dim>> nonlinear_obs_;
  Vecf<Dim> ori_, dim_;
  decimal_t start_t_ = 0;
  bool has_map_ = false;
  mplx_result res_ = mplx_result();
  mplx_plpa *lpa_ = nullptr;  // the LPA* state space of this planner (setLPAstar(true)), created by its first plan()
  int32_t lpa_control_ = MPLX_ACC;
};

typedef PolyMapPlanner<2> PolyMapPlanner2D;
typedef PolyMapPlanner<3> PolyMapPlanner3D;

}  // namespace MPL
#endif
