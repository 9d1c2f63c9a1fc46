/**
 * @file state_space.h  (mplx shim of <mpl_planner/common/state_space.h>)
 *
 * MPL::State<Coord> / MPL::StateSpace<Dim, Coord>: the search graph GraphSearch builds.  In-tree code reads
 * ss_ptr_->hm_ as (key, shared_ptr<node>) pairs with node fields pred_coord / pred_action_id / pred_action_cost
 * (poly_map_planner.h:70-86) and calls increaseCost / decreaseCost (:91-92) and getSubStateSpace
 * (map_replanner_node.cpp:245) for LPA*-style replanning.  With this back-end the graph lives in HBM; a host
 * mirror is filled on request by MapPlanner::getStateSpace() from the device dumps (mplx_result_nodes /
 * _edges / _blocked).  Incremental replanning on the voxel / occupancy map lives on the device (MPL::MapPlanner with
 * setLPAstar(true): mplx_lpa_*, its own state space in HBM); THIS host-side StateSpace is what planners with a host
 * environment hold (PolyMapPlanner::updateNodes, poly_map_planner.h:61-93): its increaseCost / decreaseCost /
 * getSubStateSpace exist, print an error and change nothing -- there is no CPU search here to repair.
 */
#ifndef MPLX_SHIM_STATE_SPACE_H
#define MPLX_SHIM_STATE_SPACE_H
#include <mpl_planner/common/env_base.h>

#include <unordered_map>

namespace MPL {

template <typename Coord>
struct State {
  State(const Coord &c) : coord(c) {}
  Coord coord;
  vec_E<Coord> succ_coord;
  std::vector<int> succ_action_id;
  std::vector<decimal_t> succ_action_cost;
  vec_E<Coord> pred_coord;
  std::vector<int> pred_action_id;
  std::vector<decimal_t> pred_action_cost;
  decimal_t g = std::numeric_limits<decimal_t>::infinity();
  decimal_t rhs = std::numeric_limits<decimal_t>::infinity();
  decimal_t h = std::numeric_limits<decimal_t>::infinity();
  bool iterationopened = false;
  bool iterationclosed = false;
};
template <typename Coord>
using StatePtr = std::shared_ptr<State<Coord>>;

template <int Dim, typename Coord>
struct StateSpace {
  StateSpace(decimal_t eps = 1) : eps_(eps) {}
  template <typename C> struct Hash { std::size_t operator()(const C &c) const { return hash_value(c); } };
  std::unordered_map<Coord, StatePtr<Coord>, Hash<Coord>> hm_;
  decimal_t eps_;
  decimal_t dt_ = 1.0;
  vec_E<StatePtr<Coord>> best_child_;
  bool need_to_reset_goal_ = false;

  void increaseCost(std::vector<std::pair<Coord, int>>) { unsupported("increaseCost"); }
  void decreaseCost(std::vector<std::pair<Coord, int>>, const std::shared_ptr<env_base<Dim>> &) { unsupported("decreaseCost"); }
  void getSubStateSpace(int) { unsupported("getSubStateSpace"); }
  void updateNode(StatePtr<Coord> &) { unsupported("updateNode"); }
  void checkValidation(const std::unordered_map<Coord, StatePtr<Coord>, Hash<Coord>> &) {}

 private:
  static void unsupported(const char *what) {
    printf(ANSI_COLOR_RED "[StateSpace] %s: this host-side state space has no search behind it (LPA* runs on the device: MPL::MapPlanner with setLPAstar(true)); nothing was changed\n" ANSI_COLOR_RESET, what);
  }
};

}  // namespace MPL
#endif
