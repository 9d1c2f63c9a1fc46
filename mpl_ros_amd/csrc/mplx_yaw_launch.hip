// mplx_yaw_launch.hip -- the YAW builds of the speculative A* kernel (mplx_spec.h): yaw-carrying states (use_yaw lattices,
// map_planner_node.cpp:119-139,165) for the ACC / JRK lattices the speculative kernel exists for.  A translation unit of its
// own so that the device code of libmplx.so keeps building in parallel.
#include <hip/hip_runtime.h>

#include "mplx_spec.h"

using namespace mplx;

// false: no yaw build for the configuration (VEL / SNP states, lattices above 128 inputs, an auxiliary map): the caller
// launches the one-node kernel astar_kernel<..., YAW>
bool mplx_launch_spec_yaw(int grid, hipStream_t s, const SearchParams &P) {
  if (!(P.control == CTRL_ACC || P.control == CTRL_JRK) || P.n_u > 128 || P.map.aux) return false;
  if (P.n_u <= 32) {  // 27 inputs: the 2-D yaw lattice of map_planner_node.cpp (3 x 3 x 3 yaw rates)
    if (P.control == CTRL_ACC)
      hipLaunchKernelGGL((astar_spec_kernel<32, 16, CTRL_ACC, 1024, 1024, false, false, true>), dim3(grid), dim3(512), 0, s, P);
    else
      hipLaunchKernelGGL((astar_spec_kernel<32, 16, CTRL_JRK, 1024, 1024, false, false, true>), dim3(grid), dim3(512), 0, s, P);
    return true;
  }
  if (P.n_u <= 64) {
    if (P.control == CTRL_ACC)
      hipLaunchKernelGGL((astar_spec_kernel<64, 4, CTRL_ACC, 512, 512, false, false, true>), dim3(grid), dim3(256), 0, s, P);
    else
      hipLaunchKernelGGL((astar_spec_kernel<64, 4, CTRL_JRK, 512, 512, false, false, true>), dim3(grid), dim3(256), 0, s, P);
    return true;
  }
  // 81 inputs: the 3-D yaw lattice (3 x 3 x 3 x 3 yaw rates)
  if (P.control == CTRL_ACC)
    hipLaunchKernelGGL((astar_spec_kernel<128, 4, CTRL_ACC, 1024, 1024, false, false, true>), dim3(grid), dim3(512), 0, s, P);
  else
    hipLaunchKernelGGL((astar_spec_kernel<128, 4, CTRL_JRK, 1024, 1024, false, false, true>), dim3(grid), dim3(512), 0, s, P);
  return true;
}
