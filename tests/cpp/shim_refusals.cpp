// The search-region / potential-field setters of the reference's MapPlanner exist on the shim and make plan() refuse
// (they would change the plan).  No GPU needed: plan() must fail before it reaches the device.
#include <mpl_planner/planner/map_planner.h>
#include <cstdio>
int main() {
  MPL::VoxelMapPlanner planner(false);
  planner.setPotentialRadius(Vec3f(1, 1, 1));
  planner.setPotentialWeight(0.1);
  planner.setGradientWeight(0.0);
  planner.setSearchRadius(Vec3f(0.5, 0.5, 0.5));
  Waypoint3D s(Control::ACC), g(Control::ACC);
  const bool ok = planner.plan(s, g);
  printf("planned %d\n", ok ? 1 : 0);
  return ok ? 1 : 0;
}
