// The reference driver mpl_test_node/src/map_planner_node.cpp:63-214 without ROS: same calls against
// the mplx shim headers (read map, setMap, freeUnknown, control set with the accumulate-by-du loops,
// start/goal Waypoint3D via use_* flags, planner setters, plan(), getTraj(), getCloseSet()).
// usage: map_planner_driver <map.bin> dx dy dz ox oy oz res sx sy sz svx svy svz gx gy gz [use_3d yaw_max u_yaw use_yaw]
// The optional tail carries the launch-file parameters of the same names (launch/map_planner_node/test.launch:27-33 passes
// yaw_max = 0.5, u_yaw = 0.5, use_3d = false, use_yaw = false; test.launch.skir: use_3d = true, yaw_max default -1).
// Prints one JSON line that tests/test_cpp_shim.py compares with the oracle.
#include <mpl_planner/planner/map_planner.h>
#include <planning_ros_utils/voxel_grid.h>

#include <cmath>
#include <cstdlib>
#include <fstream>

int main(int argc, char **argv) {
  if (argc < 18) { printf("usage\n"); return 2; }
  const int dx = atoi(argv[2]), dy = atoi(argv[3]), dz = atoi(argv[4]);
  const Vec3f ori(atof(argv[5]), atof(argv[6]), atof(argv[7]));
  const decimal_t res = atof(argv[8]);
  std::vector<signed char> map((size_t)dx * dy * dz);
  std::ifstream f(argv[1], std::ios::binary);
  f.read((char *)map.data(), map.size());

  // Initialize map util
  std::shared_ptr<MPL::VoxelMapUtil> map_util(new MPL::VoxelMapUtil);
  try {
    map_util->setMap(ori, Vec3i(dx, dy, dz), map, res);
  } catch (const std::exception &e) {
    printf("{\"error\": \"%s\"}\n", e.what());
    return 3;
  }
  // Free unknown space
  map_util->freeUnknown();
  // Inflate obstacle using robot radius (>0)  -- map_planner_node.cpp:72-85
  double robot_r = 0.0;
  if (robot_r > 0) {
    vec_Vec3i ns;
    int rn = std::ceil(robot_r / map_util->getRes());
    for (int nx = -rn; nx <= rn; nx++) {
      for (int ny = -rn; ny <= rn; ny++) {
        if (nx == 0 && ny == 0) continue;
        if (std::hypot(nx, ny) > rn) continue;
        ns.push_back(Vec3i(nx, ny, 0));
      }
    }
    map_util->dilate(ns);
  }

  // Initialize planner
  double dt = 1.0, v_max = 2.0, a_max = 1.0, u = 1.0;
  int num = 1;
  // nh.param defaults of map_planner_node.cpp:97-105, overridden like a launch file would
  bool use_3d = true, use_yaw = false;
  double yaw_max = -1.0, u_yaw = 0.3;
  if (argc >= 22) { use_3d = atoi(argv[18]) != 0; yaw_max = atof(argv[19]); u_yaw = atof(argv[20]); use_yaw = atoi(argv[21]) != 0; }
  // Set control input  (map_planner_node.cpp:107-140: the four lattices, accumulate-by-du loops)
  vec_E<VecDf> U;
  const decimal_t du = u / num;
  if (use_3d && !use_yaw) {
    for (decimal_t ddx = -u; ddx <= u; ddx += du)
      for (decimal_t ddy = -u; ddy <= u; ddy += du)
        for (decimal_t ddz = -u; ddz <= u; ddz += du) U.push_back(Vec3f(ddx, ddy, ddz));
  } else if (!use_3d && !use_yaw) {
    for (decimal_t ddx = -u; ddx <= u; ddx += du)
      for (decimal_t ddy = -u; ddy <= u; ddy += du) U.push_back(Vec3f(ddx, ddy, 0));
  } else if (!use_3d && use_yaw) {
    for (decimal_t ddx = -u; ddx <= u; ddx += du)
      for (decimal_t ddy = -u; ddy <= u; ddy += du)
        for (decimal_t dyaw = -u_yaw; dyaw <= u_yaw; dyaw += u_yaw) {
          Vec4f vec;
          vec << ddx, ddy, 0, dyaw;
          U.push_back(vec);
        }
  } else {
    for (decimal_t ddx = -u; ddx <= u; ddx += du)
      for (decimal_t ddy = -u; ddy <= u; ddy += du)
        for (decimal_t ddz = -u; ddz <= u; ddz += du)
          for (decimal_t dyaw = -u_yaw; dyaw <= u_yaw; dyaw += u_yaw) {
            Vec4f vec;
            vec << ddx, ddy, ddz, dyaw;
            U.push_back(vec);
          }
  }

  // Set start and goal
  Waypoint3D start;
  start.pos = Vec3f(atof(argv[9]), atof(argv[10]), atof(argv[11]));
  start.vel = Vec3f(atof(argv[12]), atof(argv[13]), atof(argv[14]));
  start.acc = Vec3f(0, 0, 0);
  start.jrk = Vec3f(0, 0, 0);
  start.yaw = 0;
  start.use_pos = true;
  start.use_vel = true;
  start.use_acc = false;
  start.use_jrk = false;
  start.use_yaw = use_yaw;  // if true, yaw is also propogated

  Waypoint3D goal(start.control);  // initialized with the same control as start
  goal.pos = Vec3f(atof(argv[15]), atof(argv[16]), atof(argv[17]));
  goal.vel = Vec3f(0, 0, 0);
  goal.acc = Vec3f(0, 0, 0);
  goal.jrk = Vec3f(0, 0, 0);

  std::unique_ptr<MPL::VoxelMapPlanner> planner_ptr;
  planner_ptr.reset(new MPL::VoxelMapPlanner(false));
  planner_ptr->setMapUtil(map_util);  // Set collision checking function
  planner_ptr->setVmax(v_max);        // Set max velocity
  planner_ptr->setAmax(a_max);        // Set max acceleration (as control input)
  planner_ptr->setYawmax(yaw_max);    // Set yaw threshold
  planner_ptr->setDt(dt);             // Set dt for each primitive
  planner_ptr->setU(U);               // Set control input
  planner_ptr->setTol(0.5);           // Tolerance for goal region

  bool valid = planner_ptr->plan(start, goal);
  auto traj = planner_ptr->getTraj();
  // map_planner_node.cpp:210-214, verbatim
  printf(
      "Raw traj -- J(VEL): %f, J(ACC): %f, J(JRK): %f, J(SNP): %f, J(YAW): "
      "%f, total time: %f\n",
      traj.J(Control::VEL), traj.J(Control::ACC), traj.J(Control::JRK),
      traj.J(Control::SNP), traj.Jyaw(), traj.getTotalTime());
  // the replanner's visualisation getters (map_replanner_node.cpp:78-102)
  const size_t n_expanded_nodes = planner_ptr->getExpandedNodes().size(), n_linked = planner_ptr->getLinkedNodes().size();
  const auto all_prs = planner_ptr->getAllPrimitives();
  const size_t n_expanded_edges = planner_ptr->getExpandedEdges().size();
  double prs_end_sum = 0;
  for (const auto &pr : all_prs) prs_end_sum += pr.evaluate(pr.t()).pos(0);
  printf("{\"expanded_nodes\": %zu, \"linked\": %zu, \"all_primitives\": %zu, \"expanded_edges\": %zu, \"prs_end_sum\": %.17g}\n", n_expanded_nodes, n_linked,
         all_prs.size(), n_expanded_edges, prs_end_sum);
  printf("{\"valid\": %s, \"closed\": %zu, \"expanded\": %zu, \"cost\": %.17g, \"total_time\": %.17g, \"n_prim\": %zu, \"waypoints\": [",
         valid ? "true" : "false", planner_ptr->getCloseSet().size(), planner_ptr->getExpandedNum(),
         valid ? planner_ptr->getTrajCost() : -1.0, traj.getTotalTime(), traj.getPrimitives().size());
  auto ws = traj.getWaypoints();
  for (size_t i = 0; i < ws.size(); i++)
    printf("%s[%.17g, %.17g, %.17g, %.17g, %.17g, %.17g]", i ? ", " : "", ws[i].pos(0), ws[i].pos(1), ws[i].pos(2), ws[i].vel(0), ws[i].vel(1), ws[i].vel(2));
  printf("], \"jyaw\": %.17g, \"yaws\": [", traj.Jyaw());
  for (size_t i = 0; i < ws.size(); i++) printf("%s%.17g", i ? ", " : "", ws[i].yaw);
  // the replanner's obstacle probe (map_replanner_node.cpp:177-184): cells of the start-goal ray that are occupied
  vec_Vec3i pns = map_util->rayTrace(start.pos, goal.pos);
  size_t ray_occ = 0;
  for (const auto &pn : pns)
    if (map_util->isOccupied(pn)) ray_occ++;
  // the replanner's mapper (map_replanner_node.cpp:329-331,186-188): cloud -> VoxelGrid -> getMap -> MapUtil
  VoxelGrid voxel_mapper(ori, Vec3f(dx * res, dy * res, dz * res), (float)res);
  voxel_mapper.addCloud(map_util->getCloud());
  planning_ros_msgs::VoxelMap vm = voxel_mapper.getMap();
  size_t grid_occ = 0;
  for (signed char v : vm.data) grid_occ += v > 0;
  MPL::VoxelMapUtil map_util2;
  voxel_mapper.setMapUtil(map_util2);
  printf("], \"free_start\": %s, \"ray_cells\": %zu, \"ray_occupied\": %zu, \"cloud\": %zu, \"grid_dim\": [%d, %d, %d], \"grid_occ\": %zu, \"grid_cloud2\": %zu}\n",
         map_util->isFree(start.pos) ? "true" : "false", pns.size(), ray_occ, map_util->getCloud().size(), (int)vm.dim.x, (int)vm.dim.y, (int)vm.dim.z, grid_occ,
         map_util2.getCloud().size());
  return 0;
}
