// mplx_help_launch.hip -- the helper-assisted launch: astar_spec_kernel<..., HELP = true>, whose workgroups lead
// queries while there are any, publishing their OPEN front and picking up the look-ahead cache, and turn into
// helpers afterwards (mplx_spec.h).  Third translation unit of libmplx.so (the device code builds in parallel).
#include <hip/hip_runtime.h>

#include "mplx_spec.h"

using namespace mplx;

// Lattices of at most 31 inputs (the masks of a cache record are one word): <32 lanes x 16 units>; the 125-input
// jerk lattice of BASELINE config 3 (65..128 inputs, JRK): <128 lanes x 4 units>, masks in the row.
// Returns false when no helper-capable variant exists for the configuration.
bool mplx_launch_spec_help(int grid, hipStream_t s, const SearchParams &P) {
  if (!(P.control == CTRL_ACC || P.control == CTRL_JRK) || !P.boxes) return false;
#ifdef MPLX_ONLY_ACC  // (A/B builds of the 27-input ACC kernel only: tools/build_kernel_variant.sh)
  if (P.control != CTRL_ACC || P.n_u > 31) return false;
  hipLaunchKernelGGL((astar_spec_kernel<32, 16, CTRL_ACC, 1024, 1024, true>), dim3(grid), dim3(512), 0, s, P);
  return true;
#else
  if (P.control == CTRL_JRK && P.n_u > 64 && P.n_u <= 128) {
    hipLaunchKernelGGL((astar_spec_kernel<128, 4, CTRL_JRK, 1024, 1024, true>), dim3(grid), dim3(512), 0, s, P);
    return true;
  }
  if (P.n_u > 31) return false;
  if (P.control == CTRL_ACC)
    hipLaunchKernelGGL((astar_spec_kernel<32, 16, CTRL_ACC, 1024, 1024, true>), dim3(grid), dim3(512), 0, s, P);
  else
    hipLaunchKernelGGL((astar_spec_kernel<32, 16, CTRL_JRK, 1024, 1024, true>), dim3(grid), dim3(512), 0, s, P);
  return true;
#endif
}
