/**
 * @file map_planner.h  (mplx shim of <mpl_planner/planner/map_planner.h> + planner_base.h)
 * MPL::MapPlanner<Dim> = PlannerBase<Dim, Waypoint<Dim>> for voxel / occupancy maps, with the search
 * done by libmplx.so on the GPU.  Method names, argument meaning and error behaviour follow the
 * in-tree call sites: setters store parameters, plan() returns false and
 * prints a diagnostic when the start is occupied or no trajectory is found, results are returned by
 * value.  setLPAstar(true): plan() repairs and re-uses a device-resident state space of this planner's own
 * (mplx_lpa_*), updateBlockedNodes / updateClearedNodes / getSubStateSpace / initialized as map_replanner_node.cpp uses
 * them.  Search region and potential-field cost (setSearchRegion / updatePotentialMap ...) as distance_map_planner_node.cpp
 * uses them.  Not covered by this back-end: a non-zero gradient weight, time-keyed states -- they fail loudly instead of silently
 * planning something else.
 */
#ifndef MPLX_SHIM_MAP_PLANNER_H
#define MPLX_SHIM_MAP_PLANNER_H
#include <mpl_basis/trajectory.h>
#include <mpl_collision/map_util.h>
#include <mpl_planner/common/planner_base.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cstdint>

namespace MPL {

/// ids of the planner objects of this process, never reused (the owner tag of a context's auxiliary map); the Python
/// wrapper's ids have bit 63 set
inline uint64_t next_planner_id() {
  static std::atomic<uint64_t> n{1};
  return n.fetch_add(1);
}

/// env_map<Dim>: the voxel / occupancy-map environment.  Its expansion IS the device's: get_succ runs the
/// get_succ kernel for the one node (mplx_expand_batch), is_free asks the device map.
template <int Dim>
class env_map : public env_base<Dim> {
 public:
  env_map(const std::shared_ptr<MapUtil<Dim>> &map_util) : map_util_(map_util) {}
  bool is_free(const Vecf<Dim> &pt) const override { return map_util_->isFree(map_util_->floatToInt(pt)); }
  void get_succ(const Waypoint<Dim> &curr, vec_E<Waypoint<Dim>> &succ, std::vector<decimal_t> &succ_cost, std::vector<int> &action_idx) const override {
    succ.clear(); succ_cost.clear(); action_idx.clear();
    this->expanded_nodes_.push_back(curr.pos);
    const int n_u = (int)this->U_.size();
    if (n_u == 0) return;
    mplx_waypoint c = mplx_waypoint();
    for (int i = 0; i < Dim; i++) { c.pos[i] = curr.pos(i); c.vel[i] = curr.vel(i); c.acc[i] = curr.acc(i); c.jrk[i] = curr.jrk(i); }
    c.t = curr.t;
    c.yaw = curr.yaw;                           // (read by the device when the context is configured with MPLX_YAW)
    c.control = (int32_t)curr.control & 31;
    std::vector<mplx_succ> out((size_t)n_u);
    if (mplx_expand_batch(map_util_->ctx(), 1, &c, out.data()) != MPLX_OK) {  // needs the planner set-up on the context: MapPlanner::plan / configure
      printf(ANSI_COLOR_RED "[env_map] %s\n" ANSI_COLOR_RESET, mplx_last_error(map_util_->ctx()));
      return;
    }
    for (int i = 0; i < n_u; i++) {
      if (!out[(size_t)i].valid) continue;
      Waypoint<Dim> tn(curr.control);
      for (int k = 0; k < Dim; k++) { tn.pos(k) = out[i].wp.pos[k]; tn.vel(k) = out[i].wp.vel[k]; tn.acc(k) = out[i].wp.acc[k]; tn.jrk(k) = out[i].wp.jrk[k]; }
      tn.t = out[i].wp.t;
      tn.yaw = out[i].wp.yaw;
      succ.push_back(tn);
      succ_cost.push_back(out[i].cost);
      action_idx.push_back(i);
    }
  }

 protected:
  std::shared_ptr<MapUtil<Dim>> map_util_;
};

template <int Dim>
class MapPlanner : public PlannerBase<Dim, Waypoint<Dim>> {
  typedef PlannerBase<Dim, Waypoint<Dim>> Base;
  using Base::planner_verbose_; using Base::traj_; using Base::traj_cost_; using Base::epsilon_; using Base::max_num_;
  using Base::v_max_; using Base::a_max_; using Base::j_max_; using Base::dt_; using Base::w_; using Base::t_max_;
  using Base::tol_pos_; using Base::tol_vel_; using Base::tol_acc_; using Base::heur_ignore_dynamics_;

 public:
  /// no HIP work here: planner objects may be constructed at static-initialisation time
  MapPlanner(bool verbose) : Base(verbose) {
    if (planner_verbose_) printf(ANSI_COLOR_CYAN "[MapPlanner] PLANNER VERBOSE ON (mplx back-end)\n" ANSI_COLOR_RESET);
  }
  ~MapPlanner() {  // (before the MapUtil / context it plans on: map_util_ is a member)
    if (lpa_) mplx_lpa_destroy(lpa_);
    if (map_util_ && map_util_->has_ctx()) {  // this planner's cost terms must not outlive it on the shared context
      uint64_t token = 0;
      if (mplx_aux_token(map_util_->ctx(), 0, 0, &token) == MPLX_OK && token == id_) mplx_potential_clear(map_util_->ctx());
    }
  }
  MapPlanner(const MapPlanner &) = delete;
  MapPlanner &operator=(const MapPlanner &) = delete;
  void setMapUtil(const std::shared_ptr<MapUtil<Dim>> &map_util) {
    map_util_ = map_util;
    this->ENV_.reset(new env_map<Dim>(map_util));
    this->apply_to_env();
  }
  void setU(const vec_E<VecDf> &U) override {
    Base::setU(U);
    U_.clear();
    U_yaw_.clear();
    for (const auto &u : U) {
      U_.push_back(u(0));
      U_.push_back(u(1));
      U_.push_back(Dim == 3 ? u(2) : 0.0);
      if ((int)u.size() > Dim) U_yaw_.push_back(u(Dim));  // yaw rate (Vec4f inputs, map_planner_node.cpp:125-126,135-136)
    }
    if (U_yaw_.size() != U.size()) U_yaw_.clear();
  }
  /// device pools (no reference counterpart: the reference grows std containers)
  void setCapacity(int slots, uint64_t nodes, uint64_t edges, uint64_t open_log) {
    mplx_set_capacity(map_util_->ctx(), slots, nodes, edges, open_log);
    cap_[0] = nodes; cap_[1] = edges; cap_[2] = open_log;
    if (lpa_) mplx_lpa_set_capacity(lpa_, nodes, edges, open_log);
  }
  /// LPA*: a state space exists (map_replanner_node.cpp:195,232,244)
  bool initialized() { return lpa_ && mplx_lpa_initialized(lpa_); }
  void reset() { if (lpa_) mplx_lpa_reset(lpa_); traj_ = Trajectory<Dim>(); }
  /// map_replanner_node.cpp:196,233 -- after the shared MapUtil was edited (setMap): predecessor entries whose primitive is
  /// no longer free get cost inf (increaseCost) / blocked ones that are free again get their cost back (decreaseCost).
  /// Returns the primitives of the entries that changed, like upstream (the node's commented-out publishers show them).
  vec_E<Primitive<Dim>> updateBlockedNodes(const vec_Veci<Dim> &blocked_pns) { return update_nodes(blocked_pns, true); }
  vec_E<Primitive<Dim>> updateClearedNodes(const vec_Veci<Dim> &cleared_pns) { return update_nodes(cleared_pns, false); }
  /// map_replanner_node.cpp:245: re-root the LPA* state space at the time_step-th state of the last trajectory
  void getSubStateSpace(int time_step) {
    if (!lpa_ || !send_config(control_)) return;
    if (mplx_lpa_sub_state_space(lpa_, time_step) != MPLX_OK) printf(ANSI_COLOR_RED "[MapPlanner] %s\n" ANSI_COLOR_RESET, mplx_lpa_last_error(lpa_));
  }

  /// bool PlannerBase::plan(start, goal)  (map_planner_node.cpp:187)
  bool plan(const Waypoint<Dim> &start, const Waypoint<Dim> &goal) override {
    if (planner_verbose_) { start.print("Start:"); goal.print("Goal:"); }
    traj_ = Trajectory<Dim>();
    traj_cost_ = std::numeric_limits<decimal_t>::infinity();
    if (this->unsupported_ || start.enable_t || (!U_yaw_.empty() && !start.use_yaw)) {  // never a silently different search
      printf(ANSI_COLOR_RED "[MapPlanner] plan() refused: time-keyed states, a non-zero gradient weight and a yaw-rate lattice over yaw-less states are not supported by the mplx back-end\n" ANSI_COLOR_RESET);
      return false;
    }
    mplx_ctx *ctx = map_util_->ctx();
    if (!apply_aux()) return false;
    mplx_set_record(ctx, record_cap_);  // expansion order for getExpandedNodes()
    control_ = (Control::Control)((int32_t)start.control & 31);  // (bit 16 = use_yaw = MPLX_YAW)
    if (!send_config(control_)) return false;
    mplx_waypoint s = to_c(start), g = to_c(goal);
    if (this->use_lpastar_) {
      if (!lpa_) {
        if (mplx_lpa_create(ctx, &lpa_) != MPLX_OK) return false;
        if (cap_[0]) mplx_lpa_set_capacity(lpa_, cap_[0], cap_[1], cap_[2]);
      }
      mplx_lpa_set_record(lpa_, record_cap_);
      if (mplx_lpa_plan(lpa_, &s, &g, &res_) != MPLX_OK) { printf(ANSI_COLOR_RED "[MapPlanner] %s\n" ANSI_COLOR_RESET, mplx_lpa_last_error(lpa_)); return false; }
    } else {
      if (mplx_plan(ctx, &s, &g, &res_) != MPLX_OK) { printf(ANSI_COLOR_RED "[MapPlanner] %s\n" ANSI_COLOR_RESET, mplx_last_error(ctx)); return false; }
    }
    if (res_.status == MPLX_PLAN_START_OCCUPIED) { printf(ANSI_COLOR_RED "[PlannerBase] start is not free!\n" ANSI_COLOR_RESET); return false; }
    epoch_ = mplx_plan_epoch(ctx);  // the getters answer from THIS plan only (two planners may share one MapUtil)
    traj_cost_ = res_.cost;
    if (std::isinf(traj_cost_)) { printf(ANSI_COLOR_RED "[MPPlanner] Cannot find a traj!\n" ANSI_COLOR_RESET); return false; }
    if (res_.status != MPLX_PLAN_OK) {
      // e.g. MPLX_PLAN_TRAJ_TOO_LONG: goal reached, cost known, but no trajectory came back -- never "success" with an
      // empty trajectory (a replanner would execute it)
      printf(ANSI_COLOR_RED "[MPPlanner] plan() failed with status %d: the goal was reached (cost %f) but the trajectory has more primitives "
             "than the device-side recoverTraj buffer holds\n" ANSI_COLOR_RESET, res_.status, traj_cost_);
      traj_cost_ = std::numeric_limits<decimal_t>::infinity();
      return false;
    }
    std::vector<mplx_primitive> prs(res_.traj_len > 0 ? res_.traj_len : 0);
    if (res_.traj_len > 0) {
      if (lpa_mode()) mplx_lpa_result_traj(lpa_, prs.data(), nullptr, nullptr, nullptr);
      else mplx_result_traj(ctx, 0, prs.data(), nullptr, nullptr, nullptr);
    }
    vec_E<Primitive<Dim>> out;
    for (const auto &p : prs) {
      vec_E<Vec6f> cs(Dim + ((p.control & MPLX_YAW) ? 1 : 0));
      for (int ax = 0; ax < Dim; ax++)
        for (int k = 0; k < 6; k++) cs[ax](k) = p.c[ax][k];
      if (p.control & MPLX_YAW)
        for (int k = 0; k < 6; k++) cs[Dim](k) = p.cyaw[k];
      out.push_back(Primitive<Dim>(cs, p.t, (Control::Control)p.control));
    }
    traj_ = Trajectory<Dim>(out);
    return true;
  }
  /// closed set / open set / expansion count (map_planner_node.cpp:192-196, map_replanner_node.cpp:79-94)
  vec_Vecf<Dim> getCloseSet() const override { return node_set(true); }
  vec_Vecf<Dim> getOpenSet() const override { return node_set(false); }
  size_t getExpandedNum() const { return (size_t)res_.n_expanded; }
  /// positions in expansion order = env_base::expanded_nodes_ (map_replanner_node.cpp:78-79,119); the first
  /// `setExpandedRecord(cap)` expansions are kept (default 1 << 20)
  vec_Vecf<Dim> getExpandedNodes() const override {
    vec_Vecf<Dim> ps;
    std::vector<mplx_waypoint> coords;
    if (!nodes(coords)) return ps;  // (also refuses when another planner planned on the shared context since)
    std::vector<int32_t> ids((size_t)std::max<uint64_t>(1, std::min<uint64_t>(res_.n_expanded, record_cap_)));
    uint32_t n = 0;
    if ((lpa_mode() ? mplx_lpa_result_expanded(lpa_, (uint32_t)ids.size(), ids.data(), &n) : mplx_result_expanded(map_util_->ctx(), 0, (uint32_t)ids.size(), ids.data(), &n)) != MPLX_OK) return ps;
    for (uint32_t i = 0; i < n; i++) ps.push_back(pos_of(coords[(size_t)ids[i]]));
    return ps;
  }
  void setExpandedRecord(uint32_t cap) { record_cap_ = cap; }
  /// Search-region and potential-field cost (distance_map_planner_node.cpp:185-193,199,218-224,231).  The auxiliary map
  /// (potential 0..100 per voxel, voxels outside the search region) lives on the MapUtil's device context; plan() makes
  /// sure the one on the context is this planner's (two planners may share a MapUtil) -- see mplx.h.  Upstream's
  /// implementation is un-vendored: semantics P1-P3 of DESIGN.md.
  void setSearchRadius(const Vecf<Dim> &r) { search_radius_ = v3(r); }
  void setSearchRegion(const vec_Vecf<Dim> &path, bool dense = false) {
    region_pts_.clear();
    for (const auto &p : path) { const std::array<double, 3> q = v3(p); region_pts_.insert(region_pts_.end(), q.begin(), q.end()); }
    region_dense_ = dense;
    has_region_ = !path.empty();
    aux_dirty_ = true;
  }
  void setPotentialRadius(const Vecf<Dim> &r) { pot_radius_ = v3(r); }
  void setPotentialMapRange(const Vecf<Dim> &r) { pot_range_ = v3(r); }
  void setPotentialWeight(decimal_t w) { pot_weight_ = w; aux_dirty_ = true; }
  void setGradientWeight(decimal_t w) {
    if (w != 0) refuse("setGradientWeight(!= 0)");  // only 0, the value the reference passes, is supported
  }
  void updatePotentialMap(const Vecf<Dim> &pos) {
    pot_pos_ = v3(pos);
    has_pot_ = true;
    aux_dirty_ = true;
    apply_aux();
  }
  vec_Vec3f getPotentialCloud(decimal_t h_max = 1.0) {
    vec_Vec3f out;
    if (!apply_aux()) return out;
    uint64_t n = 0;
    if (mplx_aux_cloud(map_util_->ctx(), 0, nullptr, nullptr, 0, &n) != MPLX_OK || !n) return out;
    std::vector<double> pts(3 * (size_t)n);
    std::vector<int8_t> vals((size_t)n);
    if (mplx_aux_cloud(map_util_->ctx(), 0, pts.data(), vals.data(), n, &n) != MPLX_OK) return out;
    for (uint64_t k = 0; k < n; k++) out.push_back(Vec3f(pts[3 * k], pts[3 * k + 1], Dim == 2 ? h_max * (double)vals[(size_t)k] / 100.0 : pts[3 * k + 2]));
    return out;
  }
  vec_Vecf<Dim> getSearchRegion() {
    vec_Vecf<Dim> out;
    if (!apply_aux()) return out;
    uint64_t n = 0;
    if (mplx_aux_cloud(map_util_->ctx(), 1, nullptr, nullptr, 0, &n) != MPLX_OK || !n) return out;
    std::vector<double> pts(3 * (size_t)n);
    if (mplx_aux_cloud(map_util_->ctx(), 1, pts.data(), nullptr, n, &n) != MPLX_OK) return out;
    for (uint64_t k = 0; k < n; k++) {
      Vecf<Dim> p;
      for (int i = 0; i < Dim; i++) p(i) = pts[3 * k + i];
      out.push_back(p);
    }
    return out;
  }
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
  /// the primitive of every predecessor record: Primitive(parent state, U[action], dt).  getValidPrimitives(): the
  /// finite-cost records the device stores; getAllPrimitives() (poly_map_replanner_node.cpp:184,234) adds the blocked
  /// ones (cost inf upstream), re-derived on request by mplx_result_blocked; getExpandedEdges()
  /// (map_replanner_node.cpp:100-102) keeps the edges that enter an expanded (closed) node
  /// [UNVERIFIED selection rules: upstream bodies not vendored]
  vec_E<Primitive<Dim>> getValidPrimitives() const { return edge_primitives(false, false); }
  vec_E<Primitive<Dim>> getAllPrimitives() const {
    vec_E<Primitive<Dim>> prs = edge_primitives(false, true);
    uint64_t n = 0, n_all = 0;
    if (lpa_mode()) return prs;  // (the LPA* state space keeps its blocked entries as flagged predecessor entries: already in prs)
    if (!own_results() || mplx_result_blocked(map_util_->ctx(), nullptr, nullptr, 0, &n, &n_all) != MPLX_OK || n == 0) return prs;
    std::vector<int32_t> parent((size_t)n), action((size_t)n);
    if (mplx_result_blocked(map_util_->ctx(), parent.data(), action.data(), n, &n, &n_all) != MPLX_OK) return prs;
    std::vector<mplx_waypoint> coords;
    if (!nodes(coords)) return prs;
    for (size_t e = 0; e < parent.size(); e++) prs.push_back(primitive_of(coords[(size_t)parent[e]], action[e]));
    return prs;
  }
  /// hm_.size() as upstream counts it: states reached with finite cost + states only blocked primitives reach
  size_t getStateSpaceSize() const {
    uint64_t n = 0, n_all = 0;
    if (lpa_mode()) return (size_t)res_.n_nodes;
    if (!own_results()) return 0;
    return mplx_result_blocked(map_util_->ctx(), nullptr, nullptr, 0, &n, &n_all) == MPLX_OK ? (size_t)n_all : (size_t)res_.n_nodes;
  }
  vec_E<Primitive<Dim>> getExpandedEdges() const { return edge_primitives(true, false); }
  const mplx_result &getResult() const { return res_; }

 protected:
  static mplx_waypoint to_c(const Waypoint<Dim> &w) {
    mplx_waypoint c = mplx_waypoint();
    for (int i = 0; i < Dim; i++) { c.pos[i] = w.pos(i); c.vel[i] = w.vel(i); c.acc[i] = w.acc(i); c.jrk[i] = w.jrk(i); }
    c.yaw = w.yaw; c.t = w.t;
    c.control = (int32_t)w.control & 31;
    c.enable_t = w.enable_t ? 1 : 0;
    return c;
  }
  static Vecf<Dim> pos_of(const mplx_waypoint &w) {
    Vecf<Dim> p;
    for (int k = 0; k < Dim; k++) p(k) = w.pos[k];
    return p;
  }
  /// The device keeps the state space of the context's LAST plan only.  Two planners may share one MapUtil (= one
  /// context: planner_ / replan_planner_, map_replanner_node.cpp:415,427): a getter must not size its buffers from
  /// this planner's result and then read the other planner's state space.
  bool lpa_mode() const { return this->use_lpastar_ && lpa_ != nullptr; }
  bool own_results() const {
    if (lpa_mode()) return true;  // an LPA* planner's state space is its own
    if (!map_util_ || epoch_ == 0 || mplx_plan_epoch(map_util_->ctx()) != epoch_) {
      printf(ANSI_COLOR_RED "[MapPlanner] the results of this planner's last plan() are gone: another planner sharing the MapUtil planned since\n" ANSI_COLOR_RESET);
      return false;
    }
    return true;
  }
  bool nodes(std::vector<mplx_waypoint> &coords, std::vector<int32_t> *closed = nullptr) const {
    const size_t n = (size_t)res_.n_nodes;
    if (!n || !own_results()) return false;
    coords.resize(n);
    if (closed) closed->resize(n);
    if (lpa_mode()) return mplx_lpa_result_nodes(lpa_, n, coords.data(), nullptr, nullptr, nullptr, closed ? closed->data() : nullptr, nullptr, nullptr) == MPLX_OK;
    return mplx_result_nodes(map_util_->ctx(), n, coords.data(), nullptr, nullptr, closed ? closed->data() : nullptr, nullptr) == MPLX_OK;
  }
  /// blocked (may be null): per entry, its primitive is not free in the current map -- only an LPA* state space keeps
  /// such entries (flagged, cost inf); the A* state space stores finite arrivals only (all zeros)
  bool edges(std::vector<int32_t> &child, std::vector<int32_t> &parent, std::vector<int32_t> &action, std::vector<int32_t> *blocked = nullptr) const {
    const size_t n = (size_t)res_.n_edges;
    if (!own_results()) return false;
    child.resize(n ? n : 1); parent.resize(n ? n : 1); action.resize(n ? n : 1);
    if (blocked) blocked->assign(n ? n : 1, 0);
    uint64_t m = 0;
    if ((lpa_mode() ? mplx_lpa_result_edges(lpa_, child.data(), parent.data(), action.data(), blocked ? blocked->data() : nullptr, n, &m)
                    : mplx_result_edges(map_util_->ctx(), child.data(), parent.data(), action.data(), n, &m)) != MPLX_OK) return false;
    child.resize((size_t)m); parent.resize((size_t)m); action.resize((size_t)m);
    if (blocked) blocked->resize((size_t)m);
    return true;
  }
  vec_E<Primitive<Dim>> edge_primitives(bool into_closed_only, bool with_blocked) const {
    vec_E<Primitive<Dim>> prs;
    std::vector<mplx_waypoint> coords;
    std::vector<int32_t> closed, child, parent, action, blocked;
    if (!nodes(coords, &closed) || !edges(child, parent, action, &blocked)) return prs;
    for (size_t e = 0; e < child.size(); e++) {
      if (into_closed_only && !closed[(size_t)child[e]]) continue;
      if (!with_blocked && blocked[e]) continue;  // (LPA*) an entry whose primitive passes through an obstacle: cost inf
      prs.push_back(primitive_of(coords[(size_t)parent[e]], action[e]));
    }
    return prs;
  }
  Primitive<Dim> primitive_of(const mplx_waypoint &c, int32_t action) const {
    Waypoint<Dim> w(control_);
    for (int k = 0; k < Dim; k++) { w.pos(k) = c.pos[k]; w.vel(k) = c.vel[k]; w.acc(k) = c.acc[k]; w.jrk(k) = c.jrk[k]; }
    w.t = c.t;
    w.yaw = c.yaw;
    const bool yaw_in = w.use_yaw && !U_yaw_.empty();
    VecDf u(Dim + (yaw_in ? 1 : 0));
    for (int k = 0; k < Dim; k++) u(k) = U_[3 * (size_t)action + k];
    if (yaw_in) u(Dim) = U_yaw_[(size_t)action];
    return Primitive<Dim>(w, u, dt_);
  }
  vec_Vecf<Dim> node_set(bool closed_set) const {
    vec_Vecf<Dim> ps;
    const size_t n = (size_t)res_.n_nodes;
    if (!n || !own_results()) return ps;
    std::vector<mplx_waypoint> coords(n);
    std::vector<int32_t> closed(n), opened(n);
    if ((lpa_mode() ? mplx_lpa_result_nodes(lpa_, n, coords.data(), nullptr, nullptr, nullptr, closed.data(), opened.data(), nullptr)
                    : mplx_result_nodes(map_util_->ctx(), n, coords.data(), nullptr, nullptr, closed.data(), opened.data())) != MPLX_OK) return ps;
    for (size_t i = 0; i < n; i++) {
      if (closed_set ? closed[i] : (opened[i] && !closed[i])) {
        Vecf<Dim> p;
        for (int k = 0; k < Dim; k++) p(k) = coords[i].pos[k];
        ps.push_back(p);
      }
    }
    return ps;
  }
  /// the planner set-up onto the (possibly shared) context
  bool send_config(Control::Control control) {
    mplx_ctx *ctx = map_util_->ctx();
    mplx_config cfg;
    cfg.control = (int32_t)control & 31;
    cfg.U_yaw = U_yaw_.empty() ? nullptr : U_yaw_.data();
    cfg.yaw_max = this->yaw_max_;
    cfg.tol_yaw = -1;
    cfg.n_u = (int32_t)(U_.size() / 3);
    cfg.U = U_.data();
    cfg.dt = dt_; cfg.v_max = v_max_; cfg.a_max = a_max_; cfg.j_max = j_max_;
    cfg.w = w_; cfg.eps = epsilon_;
    cfg.tol_pos = tol_pos_; cfg.tol_vel = tol_vel_; cfg.tol_acc = tol_acc_;
    cfg.t_max = t_max_;
    cfg.max_expand = max_num_;
    cfg.heur_ignore_dynamics = heur_ignore_dynamics_ ? 1 : 0;
    if (mplx_planner_config(ctx, &cfg) != MPLX_OK) { printf(ANSI_COLOR_RED "[MapPlanner] %s\n" ANSI_COLOR_RESET, mplx_last_error(ctx)); return false; }
    return true;
  }
  vec_E<Primitive<Dim>> update_nodes(const vec_Veci<Dim> &pns, bool blocked) {
    vec_E<Primitive<Dim>> changed;
    if (!lpa_ || !mplx_lpa_initialized(lpa_) || pns.empty() || !send_config(control_)) return changed;
    uint64_t n_nodes = 0, n_edges = 0;
    mplx_lpa_counts(lpa_, &n_nodes, &n_edges, nullptr);
    std::vector<int32_t> before((size_t)n_edges + 1), child((size_t)n_edges + 1);
    uint64_t m0 = 0;
    mplx_lpa_result_edges(lpa_, child.data(), nullptr, nullptr, before.data(), n_edges, &m0);
    std::vector<int32_t> c(3 * pns.size(), 0);
    for (size_t k = 0; k < pns.size(); k++)
      for (int i = 0; i < Dim; i++) c[3 * k + i] = pns[k](i);
    uint64_t n_changed = 0;
    const int rc = blocked ? mplx_lpa_update_blocked(lpa_, (int)pns.size(), c.data(), &n_changed) : mplx_lpa_update_cleared(lpa_, (int)pns.size(), c.data(), &n_changed);
    if (rc != MPLX_OK) { printf(ANSI_COLOR_RED "[MapPlanner] %s\n" ANSI_COLOR_RESET, mplx_lpa_last_error(lpa_)); return changed; }
    if (!n_changed) return changed;
    // the entries whose cost changed: flags that flipped, plus (cleared) the entries that did not exist before
    mplx_lpa_counts(lpa_, &n_nodes, &n_edges, nullptr);
    std::vector<mplx_waypoint> coords((size_t)n_nodes + 1);
    std::vector<int32_t> ch((size_t)n_edges + 1), pa((size_t)n_edges + 1), ac((size_t)n_edges + 1), bl((size_t)n_edges + 1);
    uint64_t m1 = 0;
    if (mplx_lpa_result_nodes(lpa_, n_nodes, coords.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) != MPLX_OK) return changed;
    if (mplx_lpa_result_edges(lpa_, ch.data(), pa.data(), ac.data(), bl.data(), n_edges, &m1) != MPLX_OK) return changed;
    // (entries are listed per state in arrival order: an old entry keeps its rank inside its state's list)
    size_t i0 = 0;
    for (size_t e = 0; e < (size_t)m1; e++) {
      const bool existed = i0 < (size_t)m0 && child[i0] == ch[e];
      if (existed) {
        if (before[i0] != bl[e]) changed.push_back(primitive_of(coords[(size_t)pa[e]], ac[e]));
        i0++;
      } else {
        changed.push_back(primitive_of(coords[(size_t)pa[e]], ac[e]));
      }
    }
    return changed;
  }
  static std::array<double, 3> v3(const Vecf<Dim> &v) {
    std::array<double, 3> o = {0.0, 0.0, 0.0};
    for (int i = 0; i < Dim; i++) o[(size_t)i] = v(i);
    return o;
  }
  /// make the auxiliary map on the (possibly shared) context this planner's: region first, then the potential
  bool apply_aux() {
    if (!map_util_) return true;
    mplx_ctx *ctx = map_util_->ctx();
    uint64_t token = 0;
    mplx_aux_token(ctx, 0, 0, &token);
    const uint64_t me = id_;  // (never the address: a planner allocated where a dead one lived would pass for the owner)
    if (!has_region_ && !has_pot_) {
      if (token != 0 && token != me) mplx_potential_clear(ctx);  // another planner's cost terms must not leak into this plan
      return true;
    }
    if (token == me && !aux_dirty_) return true;
    if (token != me) mplx_potential_clear(ctx);
    int rc = MPLX_OK;
    if (has_region_) rc = mplx_search_region_set(ctx, (int)(region_pts_.size() / 3), region_pts_.data(), search_radius_.data(), region_dense_ ? 1 : 0);
    if (rc == MPLX_OK) rc = mplx_potential_weights(ctx, pot_weight_, 0.0);
    if (rc == MPLX_OK && has_pot_) rc = mplx_potential_update(ctx, pot_radius_.data(), pot_pos_.data(), pot_range_.data(), 1);
    if (rc != MPLX_OK) { printf(ANSI_COLOR_RED "[MapPlanner] %s\n" ANSI_COLOR_RESET, mplx_last_error(ctx)); return false; }
    mplx_aux_token(ctx, me, 1, nullptr);
    aux_dirty_ = false;
    return true;
  }
  std::array<double, 3> search_radius_{{0, 0, 0}}, pot_radius_{{0, 0, 0}}, pot_range_{{0, 0, 0}}, pot_pos_{{0, 0, 0}};
  std::vector<double> region_pts_;
  bool region_dense_ = false, has_region_ = false, has_pot_ = false, aux_dirty_ = false;
  decimal_t pot_weight_ = 0;
  std::shared_ptr<MapUtil<Dim>> map_util_;
  mplx_lpa *lpa_ = nullptr;
  uint64_t cap_[3] = {0, 0, 0};
  std::vector<double> U_, U_yaw_;
  mplx_result res_ = mplx_result();
  uint64_t epoch_ = 0;  // mplx_plan_epoch of this planner's last plan()
  Control::Control control_ = Control::ACC;
  uint32_t record_cap_ = 1u << 20;
  const uint64_t id_ = next_planner_id();  // owner tag of the auxiliary map on the shared context (mplx_aux_token)
  void refuse(const char *what) {
    printf(ANSI_COLOR_RED "[MapPlanner] %s: not supported by the mplx back-end; plan() will fail\n" ANSI_COLOR_RESET, what);
    this->unsupported_ = true;
  }
};

typedef MapPlanner<2> OccMapPlanner;
typedef MapPlanner<3> VoxelMapPlanner;

}  // namespace MPL
#endif
