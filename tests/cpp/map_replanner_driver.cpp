// The reference's replanner mpl_test_node/src/map_replanner_node.cpp without ROS: the same globals, the same callbacks
// (replanCallback :107-173, clearCloudCallback :175-204, addCloudCallback :206-241, subtreeCallback :243-255) and the same
// set-up (main :318-444) against the mplx shim headers; the scripts of launch/map_replanner_node/ (add_cloud.sh,
// clear_cloud.sh, replan.sh, subtree.sh) are the calls at the end of main.
// usage: map_replanner_driver <map.bin> dx dy dz ox oy oz res
// Prints one JSON line per replan that tests/test_cpp_shim.py compares with the oracle.
#include <mpl_planner/planner/map_planner.h>
#include <planning_ros_utils/voxel_grid.h>

#include <unistd.h>

#include <cstdlib>
#include <fstream>

using namespace MPL;

std::shared_ptr<MPL::VoxelMapUtil> map_util;
VoxelMapPlanner planner_(false);         // (global planner objects: constructed before any device exists)
VoxelMapPlanner replan_planner_(false);
std::unique_ptr<VoxelGrid> voxel_mapper_;
Waypoint3D start, goal;
bool terminated = false;

void setMap(std::shared_ptr<MPL::VoxelMapUtil> &map_util, const planning_ros_msgs::VoxelMap &msg) {
  Vec3f ori(msg.origin.x, msg.origin.y, msg.origin.z);
  Vec3i dim(msg.dim.x, msg.dim.y, msg.dim.z);
  decimal_t res = msg.resolution;
  std::vector<signed char> map = msg.data;
  map_util->setMap(ori, dim, map, res);
}

void replanCallback() {
  if (terminated) return;
  bool valid = planner_.plan(start, goal);
  const size_t a_closed = planner_.getCloseSet().size(), a_expanded = planner_.getExpandedNodes().size();
  const double a_cost = valid ? planner_.getTrajCost() : -1.0;
  if (!valid) terminated = true;
  valid = replan_planner_.plan(start, goal);
  if (!valid) terminated = true;
  auto traj = replan_planner_.getTraj();
  printf("{\"astar_valid\": %s, \"astar_cost\": %.17g, \"astar_closed\": %zu, \"astar_expanded\": %zu, \"lpa_valid\": %s, \"lpa_cost\": %.17g, "
         "\"lpa_open\": %zu, \"lpa_closed\": %zu, \"lpa_expanded\": %zu, \"lpa_linked\": %zu, \"lpa_edges\": %zu, \"n_prim\": %zu, \"J_acc\": %.17g, \"total_time\": %.17g}\n",
         a_cost >= 0 ? "true" : "false", a_cost, a_closed, a_expanded, valid ? "true" : "false", valid ? replan_planner_.getTrajCost() : -1.0,
         replan_planner_.getOpenSet().size(), replan_planner_.getCloseSet().size(), replan_planner_.getExpandedNodes().size(),
         replan_planner_.getLinkedNodes().size(), replan_planner_.getExpandedEdges().size(), traj.getPrimitives().size(), traj.J(Control::ACC), traj.getTotalTime());
}

size_t clearCloudCallback(const vec_Vec3f &pts) {
  vec_Vec3i pns = map_util->rayTrace(pts.front(), pts.back());
  vec_Vec3i new_clear;
  for (const auto &pn : pns) {
    if (map_util->isOccupied(pn)) {
      voxel_mapper_->clear(pn(0), pn(1));
      new_clear.push_back(pn);
    }
  }
  planning_ros_msgs::VoxelMap map = voxel_mapper_->getMap();
  setMap(map_util, map);
  size_t n = 0;
  if (replan_planner_.initialized()) n = replan_planner_.updateClearedNodes(new_clear).size();
  printf("{\"cleared_cells\": %zu, \"changed_primitives\": %zu}\n", new_clear.size(), n);
  return n;
}

size_t addCloudCallback(const vec_Vec3f &pts) {
  vec_Vec3i pns = map_util->rayTrace(pts.front(), pts.back());
  vec_Vec3i ns;
  for (int nx = -2; nx <= 2; nx++)
    for (int ny = -2; ny <= 2; ny++) ns.push_back(Vec3i(nx, ny, 0));
  vec_Vec3i new_obs;
  for (const auto &it : pns) {
    for (const auto &itt : ns) {
      auto pn = it + itt;
      if (map_util->isFree(pn)) {
        voxel_mapper_->fill(pn(0), pn(1));
        new_obs.push_back(pn);
      }
    }
  }
  planning_ros_msgs::VoxelMap map = voxel_mapper_->getMap();
  setMap(map_util, map);
  size_t n = 0;
  if (replan_planner_.initialized()) n = replan_planner_.updateBlockedNodes(new_obs).size();
  printf("{\"new_obs_cells\": %zu, \"changed_primitives\": %zu}\n", new_obs.size(), n);
  return n;
}

void subtreeCallback(int data) {
  if (replan_planner_.initialized())
    replan_planner_.getSubStateSpace(data);
  else
    return;
  auto ws = replan_planner_.getTraj().getWaypoints();
  if (ws.size() < 3)
    terminated = true;
  else
    start = ws[1];
}

int main(int argc, char **argv) {
  if (argc < 9) { printf("usage\n"); return 2; }
  const int dx = atoi(argv[2]), dy = atoi(argv[3]), dz = atoi(argv[4]);
  const Vec3f origin(atof(argv[5]), atof(argv[6]), atof(argv[7]));
  const double res = atof(argv[8]);
  std::vector<signed char> data((size_t)dx * dy * dz);
  std::ifstream f(argv[1], std::ios::binary);
  f.read((char *)data.data(), data.size());
  // the bag's cloud + map -> VoxelGrid -> MapUtil (map_replanner_node.cpp:321-345); the cloud here = the map's occupied voxels
  map_util.reset(new VoxelMapUtil);
  try {
    map_util->setMap(origin, Vec3i(dx, dy, dz), data, res);
  } catch (const std::exception &e) {
    printf("{\"error\": \"%s\"}\n", e.what());
    return 3;
  }
  Vec3f dim(dx * res, dy * res, dz * res);
  voxel_mapper_.reset(new VoxelGrid(origin, dim, res));
  voxel_mapper_->addCloud(map_util->getCloud());
  planning_ros_msgs::VoxelMap map = voxel_mapper_->getMap();
  setMap(map_util, map);
  map_util->freeUnknown();

  // launch/map_replanner_node/test.launch:21-38 and the nh.param defaults of :347-398
  start.pos = Vec3f(14.5, 2.4, 0.025);
  start.vel = Vec3f(0.0, 1.0, 0.0);
  start.acc = Vec3f(0, 0, 0);
  start.jrk = Vec3f::Zero();
  start.use_pos = true;
  start.use_vel = true;
  start.use_acc = false;
  start.use_jrk = false;
  start.use_yaw = false;
  goal.pos = Vec3f(4.0, 16.0, 0.025);
  goal.vel = Vec3f(0, 0, 0);
  goal.acc = Vec3f(0, 0, 0);
  goal.jrk = Vec3f(0, 0, 0);
  goal.control = start.control;
  double dt = 1.0, v_max = 2.0, a_max = 1.0, j_max = 1.0, u_max = 1.0;
  int max_num = -1, num = 1;
  vec_E<VecDf> U;
  const decimal_t du = u_max / num;
  for (decimal_t ddx = -u_max; ddx <= u_max; ddx += du)
    for (decimal_t ddy = -u_max; ddy <= u_max; ddy += du) U.push_back(Vec3f(ddx, ddy, 0));

  planner_.setMapUtil(map_util);
  planner_.setEpsilon(1.0);
  planner_.setVmax(v_max);
  planner_.setAmax(a_max);
  planner_.setJmax(j_max);
  planner_.setDt(dt);
  planner_.setMaxNum(max_num);
  planner_.setU(U);
  planner_.setTol(0.5, 1, 1);
  planner_.setLPAstar(false);

  replan_planner_.setMapUtil(map_util);
  replan_planner_.setEpsilon(1.0);
  replan_planner_.setVmax(v_max);
  replan_planner_.setAmax(a_max);
  replan_planner_.setJmax(j_max);
  replan_planner_.setDt(dt);
  replan_planner_.setMaxNum(-1);
  replan_planner_.setU(U);
  replan_planner_.setTol(0.5, 1, 1);
  replan_planner_.setLPAstar(true);

  replanCallback();
  // add_cloud.sh, replan.sh, clear_cloud.sh, replan.sh, subtree.sh, replan.sh
  addCloudCallback({Vec3f(12.55, 9.55, 0.025), Vec3f(12.55, 11.05, 0.025)});
  replanCallback();
  clearCloudCallback({Vec3f(12.75, 9.55, 0.025), Vec3f(12.65, 11.95, 0.025)});
  replanCallback();
  subtreeCallback(1);
  replanCallback();
  // the global planners and the MapUtil they share would be destroyed by exit-time destructors, in an order relative to
  // the HIP runtime's own that nobody controls: leave without them (the process ends; the driver frees the device memory)
  fflush(stdout);
  _exit(terminated ? 1 : 0);
}
