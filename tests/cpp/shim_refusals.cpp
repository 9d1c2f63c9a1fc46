// What the shim's MapPlanner does not cover must make plan() refuse, never plan something else: a non-zero gradient
// weight (the reference only ever passes 0, distance_map_planner_node.cpp:189,220).  No GPU needed: plan() must fail
// before it reaches the device.
#include <mpl_planner/planner/map_planner.h>
#include <cstdio>
int main() {
  MPL::VoxelMapPlanner planner(false);
  planner.setPotentialRadius(Vec3f(1, 1, 1));  // (stored: the device is touched by updatePotentialMap / plan only)
  planner.setPotentialWeight(0.1);
  planner.setSearchRadius(Vec3f(0.5, 0.5, 0.5));
  planner.setGradientWeight(0.0);
  planner.setGradientWeight(0.5);
  Waypoint3D s(Control::ACC), g(Control::ACC);
  const bool ok = planner.plan(s, g);
  printf("planned %d\n", ok ? 1 : 0);
  return ok ? 1 : 0;
}
