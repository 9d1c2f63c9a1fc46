// mplx_lpa_launch.hip -- instantiates and launches the LPA* kernels (mplx_lpa.h).  Fourth translation unit of
// libmplx.so (the device code builds in parallel).  Lattices of at most 128 control inputs.
#include <hip/hip_runtime.h>

#include "mplx_lpa.h"

using namespace mplx;

template <int BLOCK>
static void launch_lpa(int what, int control, hipStream_t s, const SearchParams &P, const LpaParams &A, int mode, int pass, int grid) {
#define MPLX_LPA_CASE(C)                                                                                                   \
  if (what == 0) hipLaunchKernelGGL((lpa_plan_kernel<BLOCK, C>), dim3(1), dim3(BLOCK), 0, s, P, A);                        \
  else if (what == 1) hipLaunchKernelGGL((lpa_update_kernel<BLOCK, C>), dim3(grid), dim3(BLOCK), 0, s, P, A, mode, pass); \
  else hipLaunchKernelGGL((lpa_subtree_kernel<BLOCK, C>), dim3(1), dim3(BLOCK), 0, s, P, A);
  switch (control) {
    case CTRL_VEL: MPLX_LPA_CASE(CTRL_VEL) break;
    case CTRL_ACC: MPLX_LPA_CASE(CTRL_ACC) break;
    case CTRL_JRK: MPLX_LPA_CASE(CTRL_JRK) break;
    default: MPLX_LPA_CASE(CTRL_SNP) break;
  }
#undef MPLX_LPA_CASE
}

// what: 0 ComputeShortestPath, 1 map edit (mode 0 blocked / 1 cleared; pass 0..2 of lpa_update_kernel on `grid` workgroups),
// 2 getSubStateSpace.  false: lattice too wide.
bool mplx_launch_lpa(int what, int mode, hipStream_t s, const SearchParams &P, const LpaParams &A, int pass, int grid) {
  if (P.n_u > 128) return false;
  if (P.n_u <= 64) launch_lpa<64>(what, P.control, s, P, A, mode, pass, grid);
  else launch_lpa<128>(what, P.control, s, P, A, mode, pass, grid);
  return true;
}

// import of a finished A* into the LPA* pools (mplx_lpa.h, round 5): copy + table, blocked successors, finish
bool mplx_launch_lpa_import(hipStream_t s, const SearchParams &P, const LpaParams &A, const LpaImportArgs &I, int wide) {
  if (P.n_u > 128) return false;
#define MPLX_IMP_CASE(C)                                                                                                  \
  hipLaunchKernelGGL((lpa_import_copy_kernel<C>), dim3(wide), dim3(256), 0, s, P, A, I);                                   \
  if (P.n_u <= 64) hipLaunchKernelGGL((lpa_import_blocked_kernel<64, C>), dim3(wide), dim3(64), 0, s, P, A, I);            \
  else hipLaunchKernelGGL((lpa_import_blocked_kernel<128, C>), dim3(wide), dim3(128), 0, s, P, A, I);                     \
  hipLaunchKernelGGL((lpa_import_finish_kernel<C>), dim3(1), dim3(256), 0, s, P, A, I);
  switch (P.control) {
    case CTRL_VEL: MPLX_IMP_CASE(CTRL_VEL) break;
    case CTRL_ACC: MPLX_IMP_CASE(CTRL_ACC) break;
    case CTRL_JRK: MPLX_IMP_CASE(CTRL_JRK) break;
    default: MPLX_IMP_CASE(CTRL_SNP) break;
  }
#undef MPLX_IMP_CASE
  return true;
}

