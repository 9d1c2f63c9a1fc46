// mplx_poly_lpa_host.h -- what the LPA* of the moving-obstacle planner (mplx_poly_lpa.hip, a translation unit of its own) needs to
// know of an mplx_poly handle (mplx_poly.inl): the device view of its worlds and planner set-up, its stream, its launch guard.
#pragma once
#include <hip/hip_runtime.h>

#include "mplx_device.h"

struct mplx_poly;
struct mplx_poly_view {
  mplx::PolyDev dev;        // device arrays of the committed worlds, control inputs, limits
  int32_t general;          // hyperplane equations above degree two can occur (JRK primitives / high-degree obstacle segments)
  int32_t n_worlds, device;
  hipStream_t stream;
  mplx::GuardBlock *guard;  // host-coherent launch-guard block of the handle's planner context
  double deadline_s;        // <= 0: none
  uint64_t commit_epoch;    // number of mplx_poly_commit calls so far (the worlds a kernel sees are those of the last one)
};
// MPLX_OK, or MPLX_ERR_ARG when the handle is not configured / committed (the text is in mplx_poly_last_error)
extern "C" int mplx_poly_internal_view(mplx_poly *p, mplx_poly_view *out);
