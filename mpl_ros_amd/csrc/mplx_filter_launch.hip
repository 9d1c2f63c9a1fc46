// mplx_filter_launch.hip -- the FILTER builds of the speculative A* kernel (mplx_spec.h): the Dijkstra of getSubStateSpace
// (mplx_lpa.inl), which expands only the states that had been expanded in the state space it leaves.  Own translation unit
// (the device code of libmplx.so builds in parallel); plain kernel, no helper workgroups.
#include <hip/hip_runtime.h>

#include "mplx_spec.h"

using namespace mplx;

template <int UL, int K, int BTN, int NCAP>
static void launch_filter(int control, int grid, hipStream_t s, const SearchParams &P) {
  if (control == CTRL_ACC)
    hipLaunchKernelGGL((astar_spec_kernel<UL, K, CTRL_ACC, BTN, NCAP, false, false, false, true>), dim3(grid), dim3(UL * K), 0, s, P);
  else
    hipLaunchKernelGGL((astar_spec_kernel<UL, K, CTRL_JRK, BTN, NCAP, false, false, false, true>), dim3(grid), dim3(UL * K), 0, s, P);
}

bool mplx_launch_spec_filter(int grid, hipStream_t s, const SearchParams &P) {
  if (!(P.control == CTRL_ACC || P.control == CTRL_JRK) || P.n_u > 128 || P.map.aux || !filter_view(P).table) return false;
  if (P.n_u <= 32) launch_filter<32, 16, 1024, 1024>(P.control, grid, s, P);
  else launch_filter<128, 4, 1024, 1024>(P.control, grid, s, P);
  return true;
}
