/**
 * @file traj_solver.h  (mplx shim of <mpl_traj_solver/traj_solver.h>)
 * TrajSolver<Dim>: the calls the in-tree nodes make -- TrajSolver3D(Control::JRK), setWaypoints, setDts, solve
 * (map_planner_node.cpp:224-227, distance_map_planner_node.cpp) and setPath + solve for Control::VEL / ACC / JRK / SNP
 * (traj_solver_node.cpp:40-76).  Host code over PolySolver (poly_solver.h).
 *
 * [UNVERIFIED against upstream, whose sources are absent]: the control kind selects (smoothness order, minimised
 * derivative) = VEL (0, 1), ACC (1, 2), JRK (2, 3), SNP (3, 4); setPath fixes the position of every path point, all
 * derivatives (zero) at the two ends, and allocates segment times as L-infinity distance / v with v = 1 by default
 * (setV); the SNP solve yields septics, which a Primitive (a quintic) cannot hold -- the reference node itself marks
 * that call "does not work" (traj_solver_node.cpp:67) -- so it returns an empty Trajectory and says so.  The yaw channel
 * of the result is zero.
 */
#ifndef MPLX_SHIM_TRAJ_SOLVER_H
#define MPLX_SHIM_TRAJ_SOLVER_H
#include <mpl_traj_solver/poly_solver.h>

#include <memory>

template <int Dim>
class TrajSolver {
 public:
  TrajSolver(Control::Control control, bool debug = false) : control_(control), debug_(debug) {
    const int kind = (int)control & 15;
    const unsigned int s = kind == Control::VEL ? 0 : kind == Control::ACC ? 1 : kind == Control::JRK ? 2 : 3;
    poly_solver_.reset(new PolySolver<Dim>(s, s + 1, debug));
  }
  void setWaypoints(const vec_E<Waypoint<Dim>> &ws) { waypoints_ = ws; }
  void setDts(const std::vector<decimal_t> &dts) { dts_ = dts; }
  void setV(decimal_t v) { v_ = v; }
  /// positions only: the ends at rest in every derivative the control kind carries, the points between free but for
  /// their position; time allocation from the distances
  void setPath(const vec_Vecf<Dim> &path) {
    path_ = path;
    waypoints_.clear();
    dts_.clear();
    for (size_t i = 0; i < path.size(); i++) {
      Waypoint<Dim> w(i == 0 || i + 1 == path.size() ? (Control::Control)((int)control_ & 15) : Control::VEL);
      w.pos = path[i];
      waypoints_.push_back(w);
      if (i > 0) dts_.push_back((path[i] - path[i - 1]).template lpNorm<Eigen::Infinity>() / v_);
    }
  }
  vec_E<Waypoint<Dim>> getWaypoints() const { return waypoints_; }
  std::vector<decimal_t> getDts() const { return dts_; }
  vec_Vecf<Dim> getPath() const { return path_; }

  Trajectory<Dim> solve(bool verbose = false) {
    if (waypoints_.size() != dts_.size() + 1 || waypoints_.size() < 2) {
      if (verbose || debug_) printf(ANSI_COLOR_RED "[TrajSolver] %zu waypoints need %zu segment times, got %zu\n" ANSI_COLOR_RESET, waypoints_.size(), waypoints_.size() ? waypoints_.size() - 1 : 0, dts_.size());
      return Trajectory<Dim>();
    }
    vec_E<Primitive<Dim>> prs;
    if (!poly_solver_->solve(waypoints_, dts_)) {
      printf(ANSI_COLOR_RED "[TrajSolver] the waypoints / segment times do not determine a trajectory (a segment time <= 0, or a singular system)\n" ANSI_COLOR_RESET);
      return Trajectory<Dim>();
    }
    if (!poly_solver_->toPrimitives(prs)) {
      printf(ANSI_COLOR_RED "[TrajSolver] a minimum-snap solve yields septic segments; a Primitive holds a quintic: no Trajectory\n" ANSI_COLOR_RESET);
      return Trajectory<Dim>();
    }
    return Trajectory<Dim>(prs);
  }
  const PolySolver<Dim> &polySolver() const { return *poly_solver_; }

 private:
  Control::Control control_;
  bool debug_;
  decimal_t v_{1};
  vec_E<Waypoint<Dim>> waypoints_;
  std::vector<decimal_t> dts_;
  vec_Vecf<Dim> path_;
  std::unique_ptr<PolySolver<Dim>> poly_solver_;
};
typedef TrajSolver<2> TrajSolver2D;
typedef TrajSolver<3> TrajSolver3D;
#endif
