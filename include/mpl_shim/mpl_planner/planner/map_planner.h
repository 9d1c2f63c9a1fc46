/**
 * @file map_planner.h  (mplx shim of <mpl_planner/planner/map_planner.h> + planner_base.h)
 * MPL::MapPlanner<Dim> = PlannerBase<Dim, Waypoint<Dim>> for voxel / occupancy maps, with the search
 * done by libmplx.so on the GPU.  Method names, argument meaning and error behaviour follow the
 * in-tree call sites: setters store parameters, plan() returns false and
 * prints a diagnostic when the start is occupied or no trajectory is found, results are returned by
 * value.  Not covered by this back-end: LPA* (setLPAstar / update*Nodes), potential fields, yaw.
 */
#ifndef MPLX_SHIM_MAP_PLANNER_H
#define MPLX_SHIM_MAP_PLANNER_H
#include <mpl_basis/trajectory.h>
#include <mpl_collision/map_util.h>

namespace MPL {

template <int Dim>
class MapPlanner {
 public:
  /// no HIP work here: planner objects may be constructed at static-initialisation time
  MapPlanner(bool verbose) : planner_verbose_(verbose) {
    if (planner_verbose_) printf(ANSI_COLOR_CYAN "[MapPlanner] PLANNER VERBOSE ON (mplx back-end)\n" ANSI_COLOR_RESET);
  }
  void setMapUtil(const std::shared_ptr<MapUtil<Dim>> &map_util) { map_util_ = map_util; }
  void setVmax(decimal_t v) { v_max_ = v; }
  void setAmax(decimal_t a) { a_max_ = a; }
  void setJmax(decimal_t j) { j_max_ = j; }
  void setYawmax(decimal_t) {}  // yaw is not propagated by this back-end
  void setDt(decimal_t dt) { dt_ = dt; }
  void setW(decimal_t w) { w_ = w; }
  void setEpsilon(decimal_t eps) { epsilon_ = eps; }
  void setMaxNum(int num) { max_num_ = num; }
  void setTmax(decimal_t t) { t_max_ = t; }
  void setHeurIgnoreDynamics(bool ignore) { heur_ignore_dynamics_ = ignore; }
  void setTol(decimal_t tol_pos, decimal_t tol_vel = -1, decimal_t tol_acc = -1) {
    tol_pos_ = tol_pos; tol_vel_ = tol_vel; tol_acc_ = tol_acc;
  }
  void setU(const vec_E<VecDf> &U) {
    U_.clear();
    for (const auto &u : U) {
      U_.push_back(u(0));
      U_.push_back(u(1));
      U_.push_back(Dim == 3 ? u(2) : 0.0);
    }
  }
  /// device pools (no reference counterpart: the reference grows std containers)
  void setCapacity(int slots, uint64_t nodes, uint64_t edges, uint64_t open_log) {
    mplx_set_capacity(map_util_->ctx(), slots, nodes, edges, open_log);
  }

  /// bool PlannerBase::plan(start, goal)  (map_planner_node.cpp:187)
  bool plan(const Waypoint<Dim> &start, const Waypoint<Dim> &goal) {
    if (planner_verbose_) { start.print("Start:"); goal.print("Goal:"); }
    mplx_ctx *ctx = map_util_->ctx();
    mplx_config cfg;
    cfg.control = (int32_t)start.control & 15;
    cfg.n_u = (int32_t)(U_.size() / 3);
    cfg.U = U_.data();
    cfg.dt = dt_; cfg.v_max = v_max_; cfg.a_max = a_max_; cfg.j_max = j_max_;
    cfg.w = w_; cfg.eps = epsilon_;
    cfg.tol_pos = tol_pos_; cfg.tol_vel = tol_vel_; cfg.tol_acc = tol_acc_;
    cfg.t_max = t_max_;
    cfg.max_expand = max_num_;
    cfg.heur_ignore_dynamics = heur_ignore_dynamics_ ? 1 : 0;
    traj_ = Trajectory<Dim>();
    traj_cost_ = std::numeric_limits<decimal_t>::infinity();
    if (mplx_planner_config(ctx, &cfg) != MPLX_OK) { printf(ANSI_COLOR_RED "[MapPlanner] %s\n" ANSI_COLOR_RESET, mplx_last_error(ctx)); return false; }
    mplx_waypoint s = to_c(start), g = to_c(goal);
    if (mplx_plan(ctx, &s, &g, &res_) != MPLX_OK) { printf(ANSI_COLOR_RED "[MapPlanner] %s\n" ANSI_COLOR_RESET, mplx_last_error(ctx)); return false; }
    if (res_.status == MPLX_PLAN_START_OCCUPIED) { printf(ANSI_COLOR_RED "[PlannerBase] start is not free!\n" ANSI_COLOR_RESET); return false; }
    traj_cost_ = res_.cost;
    if (std::isinf(traj_cost_)) { printf(ANSI_COLOR_RED "[MPPlanner] Cannot find a traj!\n" ANSI_COLOR_RESET); return false; }
    std::vector<mplx_primitive> prs(res_.traj_len > 0 ? res_.traj_len : 0);
    if (res_.traj_len > 0) mplx_result_traj(ctx, 0, prs.data(), nullptr, nullptr, nullptr);
    vec_E<Primitive<Dim>> out;
    for (const auto &p : prs) {
      vec_E<Vec6f> cs(Dim);
      for (int ax = 0; ax < Dim; ax++)
        for (int k = 0; k < 6; k++) cs[ax](k) = p.c[ax][k];
      out.push_back(Primitive<Dim>(cs, p.t, (Control::Control)p.control));
    }
    traj_ = Trajectory<Dim>(out);
    return true;
  }
  Trajectory<Dim> getTraj() const { return traj_; }
  decimal_t getTrajCost() const { return traj_cost_; }
  /// closed set / open set / expansion count (map_planner_node.cpp:192-196, map_replanner_node.cpp:79-94)
  vec_Vecf<Dim> getCloseSet() const { return node_set(true); }
  vec_Vecf<Dim> getOpenSet() const { return node_set(false); }
  size_t getExpandedNum() const { return (size_t)res_.n_expanded; }
  const mplx_result &getResult() const { return res_; }

 protected:
  static mplx_waypoint to_c(const Waypoint<Dim> &w) {
    mplx_waypoint c = mplx_waypoint();
    for (int i = 0; i < Dim; i++) { c.pos[i] = w.pos(i); c.vel[i] = w.vel(i); c.acc[i] = w.acc(i); c.jrk[i] = w.jrk(i); }
    c.yaw = w.yaw; c.t = w.t;
    c.control = (int32_t)w.control & 15;
    c.enable_t = w.enable_t ? 1 : 0;
    return c;
  }
  vec_Vecf<Dim> node_set(bool closed_set) const {
    vec_Vecf<Dim> ps;
    const size_t n = (size_t)res_.n_nodes;
    if (!n) return ps;
    std::vector<mplx_waypoint> coords(n);
    std::vector<int32_t> closed(n), opened(n);
    if (mplx_result_nodes(map_util_->ctx(), coords.data(), nullptr, nullptr, closed.data(), opened.data()) != MPLX_OK) return ps;
    for (size_t i = 0; i < n; i++) {
      if (closed_set ? closed[i] : (opened[i] && !closed[i])) {
        Vecf<Dim> p;
        for (int k = 0; k < Dim; k++) p(k) = coords[i].pos[k];
        ps.push_back(p);
      }
    }
    return ps;
  }
  std::shared_ptr<MapUtil<Dim>> map_util_;
  std::vector<double> U_;
  bool planner_verbose_;
  decimal_t v_max_ = -1, a_max_ = -1, j_max_ = -1, dt_ = 1.0, w_ = 10, epsilon_ = 1.0;
  decimal_t tol_pos_ = 0.5, tol_vel_ = -1, tol_acc_ = -1, t_max_ = std::numeric_limits<decimal_t>::infinity();
  int max_num_ = -1;
  bool heur_ignore_dynamics_ = false;
  Trajectory<Dim> traj_;
  decimal_t traj_cost_ = std::numeric_limits<decimal_t>::infinity();
  mplx_result res_ = mplx_result();
};

typedef MapPlanner<2> OccMapPlanner;
typedef MapPlanner<3> VoxelMapPlanner;

}  // namespace MPL
#endif
