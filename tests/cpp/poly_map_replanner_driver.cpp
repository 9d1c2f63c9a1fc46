// The flow of the reference's moving-obstacle replanner -- mpl_test_node/src/poly_map_replanner_node.cpp:123-255 (plan(), replanCallback)
// with launch/poly_map_replanner_node/test.launch's parameters -- without ROS, through the mplx back-end: include/mpl_shim precedes the
// reference's include path, so MPL::PolyMapPlanner2D is the shim's device-backed planner (setLPAstar(true): mplx_plpa_*), while
// env_poly_map.h / poly_map_util.h / simple_obstacle.h and mpl_test_node/src/obstacle_config.hpp are the reference's own, from where
// they lie.  Per replan message (one every dt): the obstacles are where they have moved to, BOTH planners get them and the start time,
// the LPA* planner runs updateNodes() and plans, the A* planner plans afresh; costs must agree; then getSubStateSpace(1) and the robot
// moves one primitive ahead.  The obstacle course is Simple2DConfig0 of that node (its data restated: the struct lives in the node's .cpp).
// A replan that fails ends the run for both planners ("terminated" of replanCallback).
// usage: poly_map_replanner_driver [replans = 12 [start_x start_y goal_x goal_y = 2 2 38 38 (test.launch)]]
// one JSON line: per replan the two outcomes, costs, expansion counts, changed primitives
#include "obstacle_config.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

struct Simple2DConfig0 : ObstacleCourse<2> {  // poly_map_replanner_node.cpp:10-101
  Simple2DConfig0() {
    Polyhedron2D rec1;
    rec1.add(Hyperplane2D(Vec2f(-1, 0), -Vec2f::UnitX()));
    rec1.add(Hyperplane2D(Vec2f(1, 0), Vec2f::UnitX()));
    rec1.add(Hyperplane2D(Vec2f(0, -1), -Vec2f::UnitY()));
    rec1.add(Hyperplane2D(Vec2f(0, 1), Vec2f::UnitY()));
    circular_obs.push_back(PolyhedronCircularObstacle2D(rec1, Vec2f(12, 13), 5, -0.5));
    circular_obs.push_back(PolyhedronCircularObstacle2D(rec1, Vec2f(10, 20), 5, -1));
    circular_obs.push_back(PolyhedronCircularObstacle2D(rec1, Vec2f(20, 20), 6, 1, 0.5));
    nonlinear_obs.push_back(PolyhedronNonlinearObstacle2D(rec1, square(Vec2f(20, 22), Vec2f(0, -1), 8, false), 0));
    nonlinear_obs.push_back(PolyhedronNonlinearObstacle2D(rec1, square(Vec2f(25, 14), Vec2f(0, 0.5), 10, false), 0));
    nonlinear_obs.push_back(PolyhedronNonlinearObstacle2D(rec1, square(Vec2f(30, 20), Vec2f(0.2, 1.5), 10, false), 0));
    nonlinear_obs.push_back(PolyhedronNonlinearObstacle2D(rec1, back_and_forth(Vec2f(30, 30), Vec2f(1, 1), 6, Control::VEL), -6));
    nonlinear_obs.push_back(PolyhedronNonlinearObstacle2D(rec1, back_and_forth(Vec2f(30, 20), Vec2f(-0.5, 1), 10, Control::VEL), -5));
    nonlinear_obs.push_back(PolyhedronNonlinearObstacle2D(rec1, back_and_forth(Vec2f(10, 3), Vec2f(-0.2, 1), 10, Control::VEL), -2));
    nonlinear_obs.push_back(PolyhedronNonlinearObstacle2D(rec1, back_and_forth(Vec2f(5, 13), Vec2f(0.8, -1), 7, Control::VEL), -3));
    update(0);
  }
  void update(decimal_t t) {
    linear_obs.clear();
    for (const auto &it : nonlinear_obs) {
      linear_obs.push_back(it.get_linear_obstacle(t));
      linear_obs.back().set_cov_v(0.2);
    }
    for (const auto &it : circular_obs) {
      linear_obs.push_back(it.get_linear_obstacle(t));
      linear_obs.back().set_cov_v(0.4);
    }
  }
  vec_E<PolyhedronCircularObstacle2D> circular_obs;
};

int main(int argc, char **argv) {
  const int replans = argc > 1 ? atoi(argv[1]) : 12;
  const double sx = argc > 5 ? atof(argv[2]) : 2.0, sy = argc > 5 ? atof(argv[3]) : 2.0, gx = argc > 5 ? atof(argv[4]) : 38.0, gy = argc > 5 ? atof(argv[5]) : 38.0;
  Simple2DConfig0 obs;
  const Vec2f origin(0.0, 0.0), dim(40.0, 40.0);
  const double dt = 1.0, v_max = 2.0, a_max = 1.0, u = 1.0;
  vec_E<VecDf> U;
  for (decimal_t dx = -u; dx <= u; dx += u)
    for (decimal_t dy = -u; dy <= u; dy += u) U.push_back(Vec2f(dx, dy));
  std::unique_ptr<MPL::PolyMapPlanner2D> astar(new MPL::PolyMapPlanner2D(false)), lpastar(new MPL::PolyMapPlanner2D(false));
  for (auto *pl : {astar.get(), lpastar.get()}) {  // poly_map_replanner_node.cpp:327-352
    pl->setMap(origin, dim);
    pl->setStaticObstacles(obs.static_obs);
    pl->setLinearObstacles(obs.linear_obs);
    pl->setVmax(v_max);
    pl->setAmax(a_max);
    pl->setDt(dt);
    pl->setU(U);
    pl->setTol(0.5, 0.1);
    // (the node leaves the heuristic at the planner's default, the dynamics-aware one -- under which this 40 m world is a search of
    //  ~ 900 000 expansions per plan; the distance heuristic keeps the replica's plans at thousands of states, the size the
    //  one-workgroup LPA* kernel is meant for.  Both planners get the same one.)
    pl->setHeurIgnoreDynamics(true);
  }
  astar->setLPAstar(false);
  lpastar->setLPAstar(true);
  Waypoint2D start, goal;
  start.pos = Vec2f(sx, sy); start.vel = Vec2f(0, 0); start.acc = Vec2f::Zero(); start.jrk = Vec2f::Zero();
  start.use_pos = true; start.use_vel = true; start.use_acc = false; start.use_jrk = false; start.use_yaw = false;
  start.enable_t = true;
  goal.control = start.control;
  goal.pos = Vec2f(gx, gy); goal.vel = Vec2f::Zero(); goal.acc = Vec2f::Zero(); goal.jrk = Vec2f::Zero();
  goal.enable_t = true;
  decimal_t start_time = 0, plan_time = 0;
  std::string rows;  // (the planners print their own messages: the JSON line is put out whole at the end)
  int done = 0, agree = 0;
  for (int k = 0; k < replans; k++) {
    // replanCallback: plan_time += msg->data (the first message plans at plan_time 0: start_time == 0)
    if (k > 0) plan_time += dt;
    obs.update(plan_time);
    bool ok[2];
    double cost[2];
    size_t nexp[2], nb = 0, nc = 0;
    MPL::PolyMapPlanner2D *pls[2] = {astar.get(), lpastar.get()};
    for (int id = 0; id < 2; id++) {  // plan(planner_ptr, id): poly_map_replanner_node.cpp:123-186
      pls[id]->setLinearObstacles(obs.linear_obs);
      pls[id]->setStartTime(start.t);
      if (id == 1) {
        pls[id]->updateNodes();
        nb = pls[id]->getBlockedPrimitives().size();
        nc = pls[id]->getClearedPrimitives().size();
      }
      ok[id] = pls[id]->plan(start, goal);
      cost[id] = pls[id]->getTrajCost();
      nexp[id] = pls[id]->getExpandedNum();
    }
    char c0[40], c1[40];  // (JSON has no inf: a failed plan's cost is null)
    snprintf(c0, sizeof(c0), std::isfinite(cost[0]) ? "%.17g" : "null", cost[0]);
    snprintf(c1, sizeof(c1), std::isfinite(cost[1]) ? "%.17g" : "null", cost[1]);
    char row[640];
    snprintf(row, sizeof(row), "%s{\"t\": %.17g, \"astar_ok\": %d, \"lpastar_ok\": %d, \"astar_cost\": %s, \"lpastar_cost\": %s, \"astar_expanded\": %zu, \"lpastar_expanded\": %zu, "
           "\"blocked_primitives\": %zu, \"cleared_primitives\": %zu}",
           k ? ", " : "", (double)start.t, ok[0] ? 1 : 0, ok[1] ? 1 : 0, c0, c1, nexp[0], nexp[1], nb, nc);
    rows += row;
    done++;
    if (!ok[0] || !ok[1]) {
      agree += (!ok[0] && !ok[1]) ? 1 : 0;  // (both give up together)
      break;
    }
    agree += cost[0] == cost[1] ? 1 : 0;
    const auto ws = lpastar->getTraj().getWaypoints();
    if (ws.size() <= 2) break;
    start = ws[1];  // poly_map_replanner_node.cpp:224-231
    start.enable_t = true;
    start_time += dt;
    start.t = start_time;
    if (lpastar->initialized()) lpastar->getSubStateSpace(1);
  }
  printf("\n{\"replans\": [%s], \"done\": %d, \"costs_agree\": %d}\n", rows.c_str(), done, agree);
  return 0;
}
