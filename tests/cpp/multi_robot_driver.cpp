// The reference's OWN multi-robot code -- mpl_test_node/src/robot.hpp and robot_team.hpp, included from where they lie,
// unchanged -- planning through the mplx back-end: include/mpl_shim precedes the reference's include path, so
// <mpl_external_planner/poly_map_planner/poly_map_planner.h> (robot.hpp:6) is the shim's device-backed MPL::PolyMapPlanner
// while env_poly_map.h / poly_map_util.h / simple_obstacle.h stay the reference's.  main() is multi_robot_node.cpp:38-105
// without the ROS publishers: Team2 with launch/multi_robot_node/test.launch's parameters (dt 0.5, v_max 2, a_max 1, u 1,
// num 1), init(), then the 0.01 s loop of update_decentralized().  Prints every robot's current trajectory at the end
// (tests/test_cpp_shim.py compares with the Python RobotTeam on the same device back-end and, through it, with the
// compiled reference environment).
// usage: multi_robot_driver [ticks = 110]
#include "robot_team.hpp"

#include <cstdio>
#include <cstdlib>

int main(int argc, char **argv) {
  const int ticks = argc > 1 ? atoi(argv[1]) : 110;
  Vec2f origin(0.0, -5.0), dim(10.0, 10.0);
  const double dt = 0.5, v_max = 2.0, a_max = 1.0, u = 1.0;
  const int num = 1;
  vec_E<VecDf> U;
  const decimal_t du = u / num;
  for (decimal_t dx = -u; dx <= u; dx += du)
    for (decimal_t dy = -u; dy <= u; dy += du) U.push_back(Vec2f(dx, dy));
  Polyhedron2D rec;
  rec.add(Hyperplane2D(Vec2f(-0.5, 0), -Vec2f::UnitX()));
  rec.add(Hyperplane2D(Vec2f(0.5, 0), Vec2f::UnitX()));
  rec.add(Hyperplane2D(Vec2f(0, -0.5), -Vec2f::UnitY()));
  rec.add(Hyperplane2D(Vec2f(0, 0.5), Vec2f::UnitY()));
  std::unique_ptr<HomogeneousRobotTeam<2>> robot_team(new Team2(0.01));
  robot_team->set_verbose(false);
  robot_team->set_v_max(v_max);
  robot_team->set_a_max(a_max);
  robot_team->set_u(U);
  robot_team->set_dt(dt);
  robot_team->set_map(origin, dim);
  robot_team->set_geometry(rec);
  robot_team->init();
  decimal_t update_t = 0.01, time = 0;
  int done = 0;
  for (int k = 0; k < ticks; k++) {
    time += update_t;
    if (!robot_team->update_decentralized(time)) {
      printf("Robot fails to plan, ABORT!\n");
      break;
    }
    done++;
  }
  printf("{\"ticks\": %d, \"robots\": [", done);
  bool first = true;
  for (auto &it : robot_team->get_robots()) {
    const auto prs = it->get_primitives();
    printf("%s{\"n\": %zu, \"segs\": [", first ? "" : ", ", prs.size());
    first = false;
    for (size_t i = 0; i < prs.size(); i++) {
      printf("%s[", i ? ", " : "");
      for (int ax = 0; ax < 2; ax++) {
        const Vec6f c = prs[i].pr(ax).coeff();
        for (int j = 0; j < 6; j++) printf("%.17g, ", c(j));
      }
      printf("%.17g]", prs[i].t());
    }
    printf("]}");
  }
  printf("]}\n");
  MPL::shared_poly_device(true);
  return 0;
}
