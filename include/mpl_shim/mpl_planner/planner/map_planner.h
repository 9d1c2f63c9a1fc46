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

#include <algorithm>

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
    mplx_set_record(ctx, record_cap_);  // expansion order for getExpandedNodes()
    control_ = (Control::Control)cfg.control;
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
  /// positions in expansion order = env_base::expanded_nodes_ (map_replanner_node.cpp:78-79,119); the first
  /// `setExpandedRecord(cap)` expansions are kept (default 1 << 20)
  vec_Vecf<Dim> getExpandedNodes() const {
    vec_Vecf<Dim> ps;
    std::vector<mplx_waypoint> coords;
    if (!nodes(coords)) return ps;
    std::vector<int32_t> ids((size_t)std::max<uint64_t>(1, std::min<uint64_t>(res_.n_expanded, record_cap_)));
    uint32_t n = 0;
    if (mplx_result_expanded(map_util_->ctx(), 0, (uint32_t)ids.size(), ids.data(), &n) != MPLX_OK) return ps;
    for (uint32_t i = 0; i < n; i++) ps.push_back(pos_of(coords[(size_t)ids[i]]));
    return ps;
  }
  void setExpandedRecord(uint32_t cap) { record_cap_ = cap; }
  /// nodes that are linked into the graph, i.e. have at least one predecessor record (map_replanner_node.cpp:94)
  /// [UNVERIFIED: the upstream body is not vendored; node-id order here, hash-map order upstream]
  vec_Vecf<Dim> getLinkedNodes() const {
    vec_Vecf<Dim> ps;
    std::vector<mplx_waypoint> coords;
    std::vector<int32_t> child, parent, action;
    if (!nodes(coords) || !edges(child, parent, action)) return ps;
    int32_t last = -1;
    for (size_t e = 0; e < child.size(); e++)
      if (child[e] != last) { ps.push_back(pos_of(coords[(size_t)child[e]])); last = child[e]; }
    return ps;
  }
  /// the primitive of every predecessor record: Primitive(parent state, U[action], dt)
  /// (getAllPrimitives poly_map_replanner_node.cpp:184,234; blocked successors are never stored here, so
  /// getValidPrimitives() is the same list; getExpandedEdges(), map_replanner_node.cpp:100-102, keeps the
  /// edges that enter an expanded (closed) node)  [UNVERIFIED selection rules: upstream bodies not vendored]
  vec_E<Primitive<Dim>> getAllPrimitives() const { return edge_primitives(false); }
  vec_E<Primitive<Dim>> getValidPrimitives() const { return edge_primitives(false); }
  vec_E<Primitive<Dim>> getExpandedEdges() const { return edge_primitives(true); }
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
  static Vecf<Dim> pos_of(const mplx_waypoint &w) {
    Vecf<Dim> p;
    for (int k = 0; k < Dim; k++) p(k) = w.pos[k];
    return p;
  }
  bool nodes(std::vector<mplx_waypoint> &coords, std::vector<int32_t> *closed = nullptr) const {
    const size_t n = (size_t)res_.n_nodes;
    if (!n) return false;
    coords.resize(n);
    if (closed) closed->resize(n);
    return mplx_result_nodes(map_util_->ctx(), coords.data(), nullptr, nullptr, closed ? closed->data() : nullptr, nullptr) == MPLX_OK;
  }
  bool edges(std::vector<int32_t> &child, std::vector<int32_t> &parent, std::vector<int32_t> &action) const {
    const size_t n = (size_t)res_.n_edges;
    child.resize(n ? n : 1); parent.resize(n ? n : 1); action.resize(n ? n : 1);
    uint64_t m = 0;
    if (mplx_result_edges(map_util_->ctx(), child.data(), parent.data(), action.data(), n, &m) != MPLX_OK) return false;
    child.resize((size_t)m); parent.resize((size_t)m); action.resize((size_t)m);
    return true;
  }
  vec_E<Primitive<Dim>> edge_primitives(bool into_closed_only) const {
    vec_E<Primitive<Dim>> prs;
    std::vector<mplx_waypoint> coords;
    std::vector<int32_t> closed, child, parent, action;
    if (!nodes(coords, &closed) || !edges(child, parent, action)) return prs;
    for (size_t e = 0; e < child.size(); e++) {
      if (into_closed_only && !closed[(size_t)child[e]]) continue;
      const mplx_waypoint &c = coords[(size_t)parent[e]];
      Waypoint<Dim> w(control_);
      for (int k = 0; k < Dim; k++) { w.pos(k) = c.pos[k]; w.vel(k) = c.vel[k]; w.acc(k) = c.acc[k]; w.jrk(k) = c.jrk[k]; }
      w.t = c.t;
      VecDf u(Dim);
      for (int k = 0; k < Dim; k++) u(k) = U_[3 * (size_t)action[e] + k];
      prs.push_back(Primitive<Dim>(w, u, dt_));
    }
    return prs;
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
  Control::Control control_ = Control::ACC;
  uint32_t record_cap_ = 1u << 20;
};

typedef MapPlanner<2> OccMapPlanner;
typedef MapPlanner<3> VoxelMapPlanner;

}  // namespace MPL
#endif
