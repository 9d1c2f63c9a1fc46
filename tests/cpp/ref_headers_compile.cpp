// Compile check of the reference's OWN in-tree planner headers against the mplx shim: the files are included from
// where they lie under the reference tree (-I <reference>/mpl_external_planner/include), unchanged.  They pull in
// <mpl_planner/common/env_base.h>, <mpl_planner/common/planner_base.h>, <mpl_basis/primitive.h>,
// <mpl_basis/trajectory.h> and <decomp_geometry/polyhedron.h> -- all of which resolve to include/mpl_shim.
// The program then exercises what compiled: a PolyMapPlanner2D is set up like the reference's multi-robot node
// does (robot.hpp:40-63), env_poly_map::get_succ is called through the env_base interface, and plan() is
// expected to REFUSE (host environment, no CPU search in this back-end).
#include <mpl_external_planner/poly_map_planner/poly_map_planner.h>

#include <cmath>
#include <cstdio>

int main() {
  MPL::PolyMapPlanner2D planner(false);
  planner.setMap(Vec2f(0, -5), Vec2f(10, 10));
  planner.setVmax(2.0);
  planner.setAmax(1.0);
  planner.setDt(0.5);
  vec_E<VecDf> U;
  for (decimal_t dx = -1; dx <= 1; dx += 1)
    for (decimal_t dy = -1; dy <= 1; dy += 1) U.push_back(Vec2f(dx, dy));
  planner.setU(U);
  // a static box and a linear obstacle (simple_obstacle.h), a square robot geometry as in multi_robot_node.cpp:65-69
  Polyhedron2D box;
  box.add(Hyperplane2D(Vec2f(-0.5, 0), -Vec2f::UnitX()));
  box.add(Hyperplane2D(Vec2f(0.5, 0), Vec2f::UnitX()));
  box.add(Hyperplane2D(Vec2f(0, -0.5), -Vec2f::UnitY()));
  box.add(Hyperplane2D(Vec2f(0, 0.5), Vec2f::UnitY()));
  vec_E<PolyhedronObstacle2D> st;
  st.push_back(PolyhedronObstacle2D(box, Vec2f(5, 0)));
  planner.setStaticObstacles(st);
  vec_E<PolyhedronLinearObstacle2D> lin;
  lin.push_back(PolyhedronLinearObstacle2D(box, Vec2f(8, 2), Vec2f(-1, 0)));
  planner.setLinearObstacles(lin);

  Waypoint2D start;
  start.pos = Vec2f(4.0, 0.0);
  start.vel = Vec2f(1.0, 0.0);
  start.use_pos = true;
  start.use_vel = true;
  Waypoint2D goal(start.control);
  goal.pos = Vec2f(9, 0);

  // the reference's environment, driven through the env_base interface
  MPL::env_poly_map<2> env;
  (void)env;
  const auto polys = planner.getPolyhedrons(0.5);
  const bool planned = planner.plan(start, goal);  // must refuse: host environment
  Primitive2D pr(start, U[5], 0.5);
  const bool hit = collide(pr, st[0]);
  const Trajectory2D traj(vec_E<Primitive2D>(1, pr));
  // TrajectoryExtractor's sampling (trajectory_extractor.hpp:8-10): N = ceil(total / dt) -> N + 1 commands
  const auto cmds = traj.sample((int)std::ceil(traj.getTotalTime() / 0.1));
  printf("{\"cmds\": %zu, \"cmd_last_y\": %.17g, \"cmd_last_t\": %.17g}\n", cmds.size(), cmds.back().pos(1), cmds.back().t);
  printf("{\"polys\": %zu, \"planned\": %d, \"collide_static\": %d, \"J_acc\": %.17g, \"J_vel\": %.17g, \"max_vel_x\": %.17g, \"valid\": %d}\n",
         polys.size(), planned ? 1 : 0, hit ? 1 : 0, traj.J(Control::ACC), traj.J(Control::VEL), pr.max_vel(0),
         validate_primitive(pr, 2.0, 1.0, -1.0) ? 1 : 0);
  return 0;
}
