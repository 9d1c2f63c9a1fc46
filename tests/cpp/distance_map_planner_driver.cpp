// The reference's mpl_test_node/src/distance_map_planner_node.cpp:103-231 without ROS: OccMapPlanner on a 2-D map -- plan;
// then search region around that path + potential field and plan again; then the potential on the whole map -- against the
// mplx shim headers.  usage: distance_map_planner_driver <map2d.bin> dx dy ox oy res
// Prints one JSON line that tests/test_cpp_shim.py compares with the oracle.
#include <mpl_planner/planner/map_planner.h>

#include <cstdlib>
#include <fstream>

int main(int argc, char **argv) {
  if (argc < 7) { printf("usage\n"); return 2; }
  const int dx = atoi(argv[2]), dy = atoi(argv[3]);
  const Vec2f ori(atof(argv[4]), atof(argv[5]));
  const decimal_t res = atof(argv[6]);
  std::vector<signed char> data((size_t)dx * dy);
  std::ifstream f(argv[1], std::ios::binary);
  f.read((char *)data.data(), data.size());

  Waypoint2D start;
  start.pos = Vec2f(14.5, 4.5);
  start.vel = Vec2f(0, 0);
  start.acc = Vec2f(0, 0);
  start.jrk = Vec2f(0, 0);
  start.yaw = 0;
  start.use_pos = true;
  start.use_vel = true;
  start.use_acc = false;
  start.use_jrk = false;
  start.use_yaw = false;
  Waypoint2D goal(start.control);
  goal.pos = Vec2f(2.4, 16.6);
  goal.vel = Vec2f(0, 0);
  goal.acc = Vec2f(0, 0);
  goal.jrk = Vec2f(0, 0);

  std::shared_ptr<MPL::OccMapUtil> map_util = std::make_shared<MPL::OccMapUtil>();
  try {
    map_util->setMap(ori, Vec2i(dx, dy), data, res);
  } catch (const std::exception &e) {
    printf("{\"error\": \"%s\"}\n", e.what());
    return 3;
  }
  map_util->freeUnknown();

  double dt = 1.0, v_max = 2.0, a_max = 1.0, u = 1.0;
  int num = 1;
  vec_E<VecDf> U;
  const decimal_t du = u / num;
  for (decimal_t ddx = -u; ddx <= u; ddx += du)
    for (decimal_t ddy = -u; ddy <= u; ddy += du) U.push_back(Vec2f(ddx, ddy));

  std::unique_ptr<MPL::OccMapPlanner> planner_ptr;
  planner_ptr.reset(new MPL::OccMapPlanner(false));
  planner_ptr->setMapUtil(map_util);
  planner_ptr->setVmax(v_max);
  planner_ptr->setAmax(a_max);
  planner_ptr->setEpsilon(1.0);
  planner_ptr->setDt(dt);
  planner_ptr->setU(U);
  planner_ptr->setTol(0.2);
  bool valid = planner_ptr->plan(start, goal);
  const double cost0 = valid ? planner_ptr->getTrajCost() : -1.0;
  const size_t closed0 = planner_ptr->getCloseSet().size();
  double cost1 = -1, cost2 = -1;
  size_t n_region = 0, n_pot = 0, closed1 = 0, closed2 = 0;
  if (valid) {
    const auto traj = planner_ptr->getTraj();
    const auto ws = traj.getWaypoints();
    vec_Vec2f path;
    for (const auto &w : ws) path.push_back(w.pos);

    planner_ptr.reset(new MPL::OccMapPlanner(false));
    planner_ptr->setMapUtil(map_util);
    planner_ptr->setVmax(v_max);
    planner_ptr->setAmax(a_max);
    planner_ptr->setEpsilon(1.0);
    planner_ptr->setDt(dt);
    planner_ptr->setU(U);
    planner_ptr->setTol(0.5);
    planner_ptr->setSearchRadius(Vec2f(0.5, 0.5));
    planner_ptr->setSearchRegion(path);
    planner_ptr->setPotentialRadius(Vec2f(1.5, 1.5));
    planner_ptr->setPotentialWeight(10);
    planner_ptr->setGradientWeight(0);
    planner_ptr->updatePotentialMap(start.pos);
    if (planner_ptr->plan(start, goal)) cost1 = planner_ptr->getTrajCost();
    closed1 = planner_ptr->getCloseSet().size();
    n_region = planner_ptr->getSearchRegion().size();

    std::shared_ptr<MPL::OccMapUtil> global_map_util = std::make_shared<MPL::OccMapUtil>();
    global_map_util->setMap(ori, Vec2i(dx, dy), data, res);
    global_map_util->freeUnknown();
    planner_ptr.reset(new MPL::OccMapPlanner(false));
    planner_ptr->setMapUtil(global_map_util);
    planner_ptr->setVmax(v_max);
    planner_ptr->setAmax(a_max);
    planner_ptr->setEpsilon(1.0);
    planner_ptr->setDt(dt);
    planner_ptr->setU(U);
    planner_ptr->setTol(0.5);
    planner_ptr->setPotentialRadius(Vec2f(1.5, 1.5));
    planner_ptr->setPotentialWeight(10);
    planner_ptr->setGradientWeight(0);
    planner_ptr->updatePotentialMap(start.pos);
    if (planner_ptr->plan(start, goal)) cost2 = planner_ptr->getTrajCost();
    closed2 = planner_ptr->getCloseSet().size();
  }
  auto potential = planner_ptr->getPotentialCloud();
  n_pot = potential.size();
  double zmax = 0;
  for (auto &it : potential) zmax = it(2) > zmax ? it(2) : zmax;
  printf("{\"cost0\": %.17g, \"closed0\": %zu, \"cost1\": %.17g, \"closed1\": %zu, \"region\": %zu, \"cost2\": %.17g, \"closed2\": %zu, \"potential_cloud\": %zu, \"zmax\": %.17g}\n",
         cost0, closed0, cost1, closed1, n_region, cost2, closed2, n_pot, zmax);
  return 0;
}
