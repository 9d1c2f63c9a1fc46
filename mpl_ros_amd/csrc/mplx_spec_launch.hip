// mplx_spec_launch.hip -- instantiates and launches the speculative A* kernels (mplx_spec.h).
// Separate translation unit so that the device code of libmplx.so builds in two parallel halves.
#include <hip/hip_runtime.h>

#include "mplx_spec.h"

using namespace mplx;

// UL lanes per expansion unit, K units, BTN batch-table slots (>= 2 K n_u), NCAP near-set capacity
template <int UL, int K, int BTN, int NCAP>
static void launch_spec(int control, int grid, hipStream_t s, const SearchParams &P) {
  if (control == CTRL_ACC)
    hipLaunchKernelGGL((astar_spec_kernel<UL, K, CTRL_ACC, BTN, NCAP>), dim3(grid), dim3(UL * K), 0, s, P);
  else
    hipLaunchKernelGGL((astar_spec_kernel<UL, K, CTRL_JRK, BTN, NCAP>), dim3(grid), dim3(UL * K), 0, s, P);
}

// speculation: -1 / >1 = widest variant for the lattice; 8, 4, 2 = narrower variants (A/B measurements)
// P.map.aux (potential field / search region): the POT builds (round 3: the <= 32-input ACC variant, the distance-map planner's
// lattices; round 4: JRK states and lattices up to 128 inputs as well)
bool mplx_launch_spec(int speculation, int grid, hipStream_t s, const SearchParams &P) {
  if (!(P.control == CTRL_ACC || P.control == CTRL_JRK) || P.n_u > 128) return false;  // built for the reference's lattices
#ifdef MPLX_ONLY_ACC  // (A/B builds of the 27-input ACC kernel only: tools/build_kernel_variant.sh)
  if (P.control != CTRL_ACC || P.n_u > 31 || P.map.aux) return false;
  hipLaunchKernelGGL((astar_spec_kernel<32, 16, CTRL_ACC, 1024, 1024>), dim3(grid), dim3(512), 0, s, P);
  return true;
#else
  if (P.map.aux) {  // POT builds: the 16-unit kernel for lattices of at most 32 inputs, four 128-lane units up to 128 inputs
    if (P.n_u <= 32) {
      if (P.control == CTRL_ACC) hipLaunchKernelGGL((astar_spec_kernel<32, 16, CTRL_ACC, 1024, 1024, false, true>), dim3(grid), dim3(512), 0, s, P);
      else hipLaunchKernelGGL((astar_spec_kernel<32, 16, CTRL_JRK, 1024, 1024, false, true>), dim3(grid), dim3(512), 0, s, P);
    } else {
      if (P.control == CTRL_ACC) hipLaunchKernelGGL((astar_spec_kernel<128, 4, CTRL_ACC, 1024, 1024, false, true>), dim3(grid), dim3(512), 0, s, P);
      else hipLaunchKernelGGL((astar_spec_kernel<128, 4, CTRL_JRK, 1024, 1024, false, true>), dim3(grid), dim3(512), 0, s, P);
    }
    return true;
  }
  if (P.n_u <= 32 && speculation == 8)
    launch_spec<64, 8, 512, 512>(P.control, grid, s, P);     // 8 units of one wave each
  else if (P.n_u <= 32)
    launch_spec<32, 16, 1024, 1024>(P.control, grid, s, P);  // 16 units, two per wave
  else if (P.n_u <= 64)
    launch_spec<64, 4, 512, 512>(P.control, grid, s, P);     // 4 units of one wave each
  else if (speculation == 2)
    launch_spec<128, 2, 512, 512>(P.control, grid, s, P);    // 2 units of two waves each
  else
    launch_spec<128, 4, 1024, 1024>(P.control, grid, s, P);  // 4 units of two waves each
  return true;
#endif
}
