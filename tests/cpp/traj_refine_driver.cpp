// The reference driver after the plan (map_planner_node.cpp:205-227): raw trajectory -> waypoints, intermediate controls
// to VEL, segment times -> TrajSolver3D(Control::JRK) -> refined trajectory, with the node's two J print-outs.  Compiled
// against the drop-in headers; with -DMPLX_WITH_REFERENCE_GLUE the reference's own planning_ros_utils glue
// (primitive_ros_utils.h, trajectory_extractor.hpp -- taken from the reference checkout, unchanged) converts the result to
// a Trajectory message and back and samples it at 100 Hz like trajectory_extractor_node.cpp.
#include <mpl_traj_solver/traj_solver.h>
#ifdef MPLX_WITH_REFERENCE_GLUE
#include <trajectory_extractor.hpp>
#endif
#include <cstdio>

int main() {
  // a raw trajectory as the ACC search returns it: five 1 s primitives from (1, 2, 0.5) at rest
  const double us[5][3] = {{1, 0, 0}, {1, 1, 0}, {0, 1, 1}, {-1, 0, 0}, {0, -1, -1}};
  Waypoint3D w(Control::ACC);
  w.pos = Vec3f(1, 2, 0.5);
  vec_E<Primitive3D> prs;
  for (int i = 0; i < 5; i++) {
    VecDf u(3);
    for (int k = 0; k < 3; k++) u(k) = us[i][k];
    prs.push_back(Primitive3D(w, u, 1.0));
    w = prs.back().evaluate(1.0);
  }
  Trajectory3D traj(prs);
  printf("Raw traj -- J(VEL): %f, J(ACC): %f, J(JRK): %f, J(SNP): %f, J(YAW): %f, total time: %f\n", traj.J(Control::VEL), traj.J(Control::ACC),
         traj.J(Control::JRK), traj.J(Control::SNP), traj.Jyaw(), traj.getTotalTime());
  const double raw_j[4] = {traj.J(Control::VEL), traj.J(Control::ACC), traj.J(Control::JRK), traj.J(Control::SNP)};
  // map_planner_node.cpp:217-227, verbatim
  auto waypoints = traj.getWaypoints();
  for (size_t i = 1; i < waypoints.size() - 1; i++)
    waypoints[i].control = Control::VEL;
  auto dts = traj.getSegmentTimes();
  TrajSolver3D traj_solver(Control::JRK);
  traj_solver.setWaypoints(waypoints);
  traj_solver.setDts(dts);
  traj = traj_solver.solve();
  printf("Refined traj -- J(VEL): %f, J(ACC): %f, J(JRK): %f, J(SNP): %f, J(YAW): %f, total time: %f\n", traj.J(Control::VEL), traj.J(Control::ACC),
         traj.J(Control::JRK), traj.J(Control::SNP), traj.Jyaw(), traj.getTotalTime());
  size_t n_cmds = 0;
  bool roundtrip = false;
  double mid[7] = {0, 0, 0, 0, 0, 0, 0};
#ifdef MPLX_WITH_REFERENCE_GLUE
  planning_ros_msgs::Trajectory msg = toTrajectoryROSMsg(traj);
  const Trajectory3D back = toTrajectory3D(msg);
  roundtrip = back.segs.size() == traj.segs.size() && back.getTotalTime() == traj.getTotalTime();
  for (size_t i = 0; roundtrip && i < back.segs.size(); i++)
    for (int k = 0; k < 3; k++) roundtrip = roundtrip && back.segs[i].pr(k).coeff() == traj.segs[i].pr(k).coeff();
  TrajectoryExtractor extractor(msg, 0.01);
  const auto cmds = extractor.getCommands();
  n_cmds = cmds.size();
  const auto &c = cmds[n_cmds / 2];
  const double m[7] = {c.header.stamp.toSec(), c.position.x, c.position.y, c.position.z, c.velocity.x, c.velocity.y, c.velocity.z};
  for (int i = 0; i < 7; i++) mid[i] = m[i];
#endif
  printf("{\"raw_J\": [%.17g, %.17g, %.17g, %.17g], \"refined_J\": [%.17g, %.17g, %.17g, %.17g], \"coeff\": [", raw_j[0], raw_j[1], raw_j[2], raw_j[3],
         traj.J(Control::VEL), traj.J(Control::ACC), traj.J(Control::JRK), traj.J(Control::SNP));
  for (size_t i = 0; i < traj.segs.size(); i++) {
    printf("%s[", i ? ", " : "");
    for (int k = 0; k < 3; k++) {
      const Vec6f c6 = traj.segs[i].pr(k).coeff();
      printf("%s[%.17g, %.17g, %.17g, %.17g, %.17g, %.17g]", k ? ", " : "", c6(0), c6(1), c6(2), c6(3), c6(4), c6(5));
    }
    printf("]");
  }
#ifdef MPLX_WITH_REFERENCE_GLUE
  const char *glue = "true";
#else
  const char *glue = "false";
#endif
  printf("], \"with_reference_glue\": %s, \"n_cmds\": %zu, \"roundtrip_equal\": %s, \"cmd_mid\": [%.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g]}\n", glue, n_cmds,
         roundtrip ? "true" : "false", mid[0], mid[1], mid[2], mid[3], mid[4], mid[5], mid[6]);
  return 0;
}
