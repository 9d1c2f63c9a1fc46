/** @file decomp_basis/data_type.h  (mplx shim): DecompUtil shares the typedefs of mpl_basis/data_type.h */
#ifndef MPLX_SHIM_DECOMP_DATA_TYPE_H
#define MPLX_SHIM_DECOMP_DATA_TYPE_H
#include <mpl_basis/data_type.h>
/// tolerance of Polyhedron::inside  [UNVERIFIED recollection of DecompUtil decomp_basis/data_type.h: epsilon_ = 1e-10]
constexpr decimal_t epsilon_ = 1e-10;
#endif
