/**
 * @file env_base.h  (mplx shim of <mpl_planner/common/env_base.h>)
 *
 * MPL::env_base<Dim>: the environment interface GraphSearch expands through.  The in-tree environments subclass
 * it (env_poly_map.h:17-73, env_cloud.h:17-74) and use: virtual is_free(pt) / is_free(pr) / get_succ(curr, succ,
 * succ_cost, action_idx) const / calculate_intrinsic_cost(pr), forward_action(curr, action_id, pr)
 * (poly_map_planner.h:76), and the members U_, dt_, v_max_, a_max_, j_max_, yaw_max_, w_ and the MUTABLE
 * expanded_nodes_ (pushed to inside a const method, env_poly_map.h:52).
 *
 * With this back-end the voxel / occupancy-map environment (env_map, map_planner.h) runs on the device; an
 * env_base subclass written in host C++ cannot, so PlannerBase::plan() refuses to search through one (see
 * planner_base.h) -- the class is here so that such code compiles and its get_succ can be called and checked.
 * Heuristic and goal test restate what the device and the oracle compute (oracle/mpl_oracle.c cal_heur).
 */
#ifndef MPLX_SHIM_ENV_BASE_H
#define MPLX_SHIM_ENV_BASE_H
#include <mpl_basis/trajectory.h>

namespace MPL {

template <int Dim>
class env_base {
 public:
  env_base() {}
  virtual ~env_base() {}

  /// goal test: position within tol_pos (L-inf), then velocity / acceleration when the goal carries them and the
  /// tolerance is >= 0; reaching t_max also ends the search  [UNVERIFIED rule, same as device / oracle]
  virtual bool is_goal(const Waypoint<Dim> &state) const {
    if (state.t >= t_max_) return true;
    bool goaled = linf(state.pos, goal_node_.pos) <= tol_pos_;
    if (goaled && goal_node_.use_vel && tol_vel_ >= 0) goaled = linf(state.vel, goal_node_.vel) <= tol_vel_;
    if (goaled && goal_node_.use_acc && tol_acc_ >= 0) goaled = linf(state.acc, goal_node_.acc) <= tol_acc_;
    return goaled;
  }
  /// 0 at the goal key; otherwise the distance-only bound of cal_heur.  (The dynamics-aware closed forms of
  /// env_base::cal_heur live on the device, mplx_heuristic_batch; host code that needs them calls that.)
  virtual decimal_t get_heur(const Waypoint<Dim> &state) const {
    if (goal_node_ == state) return 0;
    const decimal_t d = linf(state.pos, goal_node_.pos);
    return v_max_ > 0 ? w_ * d / v_max_ : w_ * d;
  }
  /// Primitive from a state and the action_id-th control input (poly_map_planner.h:76)
  void forward_action(const Waypoint<Dim> &curr, int action_id, Primitive<Dim> &pr) const { pr = Primitive<Dim>(curr, U_[action_id], dt_); }

  void set_u(const vec_E<VecDf> &U) { U_ = U; }
  void set_v_max(decimal_t v) { v_max_ = v; }
  void set_a_max(decimal_t a) { a_max_ = a; }
  void set_j_max(decimal_t j) { j_max_ = j; }
  void set_yaw_max(decimal_t yaw) { yaw_max_ = yaw; }
  void set_dt(decimal_t dt) { dt_ = dt; }
  void set_w(decimal_t w) { w_ = w; }
  void set_wyaw(decimal_t wyaw) { wyaw_ = wyaw; }
  void set_tol_pos(decimal_t pos) { tol_pos_ = pos; }
  void set_tol_vel(decimal_t vel) { tol_vel_ = vel; }
  void set_tol_acc(decimal_t acc) { tol_acc_ = acc; }
  void set_t_max(int t) { t_max_ = t; }
  void set_heur_ignore_dynamics(bool ignore) { heur_ignore_dynamics_ = ignore; }
  bool set_goal(const Waypoint<Dim> &state) { goal_node_ = state; return true; }
  virtual void set_prior_trajectory(const Trajectory<Dim> &) {}
  virtual void set_search_region(const std::vector<bool> &search_region) { search_region_ = search_region; }

  /// point / primitive collision tests and the expansion: what a concrete environment overrides
  virtual bool is_free(const Vecf<Dim> &) const { printf("Used Null is_free() for pt\n"); return true; }
  virtual bool is_free(const Primitive<Dim> &) const { printf("Used Null is_free() for pr\n"); return true; }
  virtual decimal_t calculate_intrinsic_cost(const Primitive<Dim> &pr) const { return pr.J(pr.control()) + w_ * dt_; }
  virtual void get_succ(const Waypoint<Dim> &curr, vec_E<Waypoint<Dim>> &succ, std::vector<decimal_t> &succ_cost, std::vector<int> &action_idx) const {
    (void)curr;
    printf("Used Null get_succ()\n");
    succ.clear(); succ_cost.clear(); action_idx.clear();
  }
  virtual void info() {
    printf("++++++++++++++++++++ env_base ++++++++++++++++++\n");
    printf("+  w: %.2f  dt: %.2f  v_max: %.2f  a_max: %.2f  j_max: %.2f  U num: %zu  tol_pos: %.2f\n", w_, dt_, v_max_, a_max_, j_max_, U_.size(), tol_pos_);
  }
  decimal_t get_dt() const { return dt_; }
  vec_Vecf<Dim> get_expanded_nodes() const { return expanded_nodes_; }

  // members are public upstream too: planners and environments reach into them (env_poly_map.h:54-66)
  bool heur_ignore_dynamics_{false};
  decimal_t tol_pos_{0.5}, tol_vel_{-1}, tol_acc_{-1};
  vec_E<VecDf> U_;
  decimal_t v_max_{-1}, a_max_{-1}, j_max_{-1}, yaw_max_{-1};
  decimal_t t_max_{std::numeric_limits<decimal_t>::infinity()};
  decimal_t dt_{1.0};
  decimal_t w_{10}, wyaw_{1};
  mutable vec_Vecf<Dim> expanded_nodes_;
  mutable vec_E<Primitive<Dim>> expanded_edges_;
  Waypoint<Dim> goal_node_;
  std::vector<bool> search_region_;

 protected:
  static decimal_t linf(const Vecf<Dim> &a, const Vecf<Dim> &b) { return (a - b).lpNormInf(); }
};

}  // namespace MPL
#endif
