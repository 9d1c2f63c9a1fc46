/**
 * @file planner_base.h  (mplx shim of <mpl_planner/common/planner_base.h>)
 *
 * MPL::PlannerBase<Dim, Coord>: what the in-tree planners derive from (poly_map_planner.h:18-93,
 * ellipsoid_planner.h) and what the nodes call: the setters, plan(start, goal),
 * getTraj / getTrajCost / getCloseSet / getOpenSet / getExpandedNodes / getAllPrimitives ..., setLPAstar, and the protected
 * ENV_ (std::shared_ptr<env_base<Dim>>), ss_ptr_ (std::shared_ptr<StateSpace<Dim, Coord>>), planner_verbose_.
 *
 * The search itself is the device's: MPL::MapPlanner (map_planner.h) overrides plan() with the C-ABI call.  A
 * planner whose environment is host C++ (an env_base subclass such as the reference's env_poly_map) cannot be
 * expanded on the device; PlannerBase::plan() then fails loudly instead of running a CPU search.
 */
#ifndef MPLX_SHIM_PLANNER_BASE_H
#define MPLX_SHIM_PLANNER_BASE_H
#include <mpl_planner/common/state_space.h>

namespace MPL {

template <int Dim, typename Coord>
class PlannerBase {
 public:
  PlannerBase(bool verbose = false) : planner_verbose_(verbose) {}
  virtual ~PlannerBase() {}

  /// state space exists (LPA* flow: map_replanner_node.cpp:195)
  bool initialized() { return ss_ptr_ != nullptr; }
  Trajectory<Dim> getTraj() const { return traj_; }
  decimal_t getTrajCost() const { return traj_cost_; }
  /// primitives of the pred entries of the state-space mirror: finite cost only / all (poly_map_replanner_node.cpp:184,234)
  vec_E<Primitive<Dim>> getValidPrimitives() const { return mirror_primitives(false); }
  vec_E<Primitive<Dim>> getAllPrimitives() const { return mirror_primitives(true); }
  /// positions handed to get_succ, in order (env_base::expanded_nodes_)
  virtual vec_Vecf<Dim> getExpandedNodes() const { return ENV_ ? ENV_->get_expanded_nodes() : vec_Vecf<Dim>(); }
  virtual vec_Vecf<Dim> getCloseSet() const { return mirror_set(true); }
  virtual vec_Vecf<Dim> getOpenSet() const { return mirror_set(false); }
  void getSubStateSpace(int time_step) { if (ss_ptr_) ss_ptr_->getSubStateSpace(time_step); }
  void checkValidation() { if (ss_ptr_) ss_ptr_->checkValidation(ss_ptr_->hm_); }
  void reset() { ss_ptr_ = nullptr; traj_ = Trajectory<Dim>(); }

  /// LPA* (map_replanner_node.cpp:425-437): MapPlanner::plan then repairs and re-uses its device-resident state space
  void setLPAstar(bool use_lpastar) { use_lpastar_ = use_lpastar; }
  virtual void setEpsilon(decimal_t eps) { epsilon_ = eps; }
  virtual void setVmax(decimal_t v) { v_max_ = v; if (ENV_) ENV_->set_v_max(v); }
  virtual void setAmax(decimal_t a) { a_max_ = a; if (ENV_) ENV_->set_a_max(a); }
  virtual void setJmax(decimal_t j) { j_max_ = j; if (ENV_) ENV_->set_j_max(j); }
  /// The yaw threshold (map_planner_node.cpp:179-180) constrains yaw-carrying primitives only: the reference node
  /// always calls it, and its own config-1 launch file passes yaw_max = 0.5 with use_yaw = false
  /// (launch/map_planner_node/test.launch:28,33).  It takes effect when the search states carry yaw (use_yaw).
  virtual void setYawmax(decimal_t yaw) { yaw_max_ = yaw; if (ENV_) ENV_->set_yaw_max(yaw); }
  virtual void setTmax(decimal_t t) { t_max_ = t; if (ENV_) ENV_->t_max_ = t; }
  virtual void setDt(decimal_t dt) { dt_ = dt; if (ENV_) ENV_->set_dt(dt); }
  virtual void setW(decimal_t w) { w_ = w; if (ENV_) ENV_->set_w(w); }
  virtual void setMaxNum(int num) { max_num_ = num; }
  virtual void setU(const vec_E<VecDf> &U) {
    U_vec_ = U;  // (Dim + 1 components: the last one is the yaw rate of the use_yaw lattices, map_planner_node.cpp:119-139)
    if (ENV_) ENV_->set_u(U);
  }
  virtual void setTol(decimal_t tol_pos, decimal_t tol_vel = -1, decimal_t tol_acc = -1) {
    tol_pos_ = tol_pos; tol_vel_ = tol_vel; tol_acc_ = tol_acc;
    if (ENV_) { ENV_->set_tol_pos(tol_pos); ENV_->set_tol_vel(tol_vel); ENV_->set_tol_acc(tol_acc); }
  }
  virtual void setHeurIgnoreDynamics(bool ignore) { heur_ignore_dynamics_ = ignore; if (ENV_) ENV_->set_heur_ignore_dynamics(ignore); }
  virtual void setPriorTrajectory(const Trajectory<Dim> &traj) { if (ENV_) ENV_->set_prior_trajectory(traj); }

  /// The device plans through MapPlanner::plan.  Searching through an arbitrary host environment is not something
  /// this back-end can do (and it has no CPU search to fall back to): say so and fail.
  virtual bool plan(const Coord &start, const Coord &goal) {
    (void)start; (void)goal;
    printf(ANSI_COLOR_RED "[PlannerBase] plan(): this planner's environment is host code (an env_base subclass); the mplx back-end "
           "expands on the device only (MPL::MapPlanner) and has no CPU search -- not planned\n" ANSI_COLOR_RESET);
    traj_ = Trajectory<Dim>();
    traj_cost_ = std::numeric_limits<decimal_t>::infinity();
    return false;
  }

 protected:
  /// push the stored set-up into a freshly created environment (setMapUtil-style calls may come after the setters)
  void apply_to_env() {
    if (!ENV_) return;
    ENV_->set_v_max(v_max_); ENV_->set_a_max(a_max_); ENV_->set_j_max(j_max_); ENV_->set_yaw_max(yaw_max_);
    ENV_->set_dt(dt_); ENV_->set_w(w_); ENV_->t_max_ = t_max_;
    ENV_->set_tol_pos(tol_pos_); ENV_->set_tol_vel(tol_vel_); ENV_->set_tol_acc(tol_acc_);
    ENV_->set_heur_ignore_dynamics(heur_ignore_dynamics_);
    if (!U_vec_.empty()) ENV_->set_u(U_vec_);
  }
  vec_E<Primitive<Dim>> mirror_primitives(bool all) const {
    vec_E<Primitive<Dim>> prs;
    if (!ss_ptr_ || !ENV_) return prs;
    for (const auto &it : ss_ptr_->hm_)
      for (size_t i = 0; i < it.second->pred_coord.size(); i++) {
        if (!all && std::isinf(it.second->pred_action_cost[i])) continue;
        Primitive<Dim> pr;
        ENV_->forward_action(it.second->pred_coord[i], it.second->pred_action_id[i], pr);
        prs.push_back(pr);
      }
    return prs;
  }
  vec_Vecf<Dim> mirror_set(bool closed) const {
    vec_Vecf<Dim> ps;
    if (!ss_ptr_) return ps;
    for (const auto &it : ss_ptr_->hm_)
      if (closed ? it.second->iterationclosed : (it.second->iterationopened && !it.second->iterationclosed)) ps.push_back(it.second->coord.pos);
    return ps;
  }

  std::shared_ptr<env_base<Dim>> ENV_;
  std::shared_ptr<StateSpace<Dim, Coord>> ss_ptr_;
  Trajectory<Dim> traj_;
  decimal_t traj_cost_ = std::numeric_limits<decimal_t>::infinity();
  decimal_t epsilon_ = 1.0;
  int max_num_ = -1;
  bool planner_verbose_ = false;
  bool use_lpastar_ = false;
  bool unsupported_ = false;  // a request this back-end does not cover was made: plan() must fail
  // the set-up as the setters received it (re-applied when an environment is created later)
  decimal_t v_max_ = -1, a_max_ = -1, j_max_ = -1, yaw_max_ = -1, dt_ = 1.0, w_ = 10;
  decimal_t tol_pos_ = 0.5, tol_vel_ = -1, tol_acc_ = -1, t_max_ = std::numeric_limits<decimal_t>::infinity();
  bool heur_ignore_dynamics_ = false;
  vec_E<VecDf> U_vec_;
};

}  // namespace MPL
#endif
