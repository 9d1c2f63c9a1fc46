// mplx_poly_launch.hip -- instantiates and launches the kernels of the moving-obstacle environment (mplx_poly_search.h):
// env_poly_map::get_succ for a batch of nodes and the device-resident search.  Its own translation unit (the fifth with
// device code) so that libmplx.so builds in parallel.
#include <hip/hip_runtime.h>

#include "mplx_poly_search.h"

using namespace mplx;

// general: hyperplane equations above degree two can occur (JRK / SNP primitives, obstacle trajectories with such segments)
bool mplx_launch_poly_get_succ(bool general, int grid, hipStream_t s, const PolyDev &D, int K, const int32_t *world_of, const double *states, PolySuccOut *out, int32_t *flags) {
  if (general)
    hipLaunchKernelGGL((poly_get_succ_kernel<256, true>), dim3(grid), dim3(256), 0, s, D, K, world_of, states, out, flags);
  else
    hipLaunchKernelGGL((poly_get_succ_kernel<256, false>), dim3(grid), dim3(256), 0, s, D, K, world_of, states, out, flags);
  return true;
}
// control: ACC or JRK states (the caller has refused anything else); block: 256 lanes (a search that runs its own
// collision tests, and the helper workgroups), or 64 (a leader whose collision tests are served by helpers: what is left
// on it -- pop, nine primitives, commit -- needs one wave, and one wave has no cross-wave reductions or barrier waits)
template <int BLOCK>
static void launch_poly(int control, bool general, int grid, hipStream_t s, const SearchParams &P) {
  if (control == CTRL_JRK)
    hipLaunchKernelGGL((astar_poly_kernel<BLOCK, CTRL_JRK, true>), dim3(grid), dim3(BLOCK), 0, s, P);
  else if (general)
    hipLaunchKernelGGL((astar_poly_kernel<BLOCK, CTRL_ACC, true>), dim3(grid), dim3(BLOCK), 0, s, P);
  else
    hipLaunchKernelGGL((astar_poly_kernel<BLOCK, CTRL_ACC, false>), dim3(grid), dim3(BLOCK), 0, s, P);
}
bool mplx_launch_poly_search(int control, bool general, int block, int grid, hipStream_t s, const SearchParams &P) {
  if (block == 64) launch_poly<64>(control, general, grid, s, P);
  else launch_poly<256>(control, general, grid, s, P);
  return true;
}
