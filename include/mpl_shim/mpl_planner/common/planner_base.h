/** @file planner_base.h  (mplx shim of <mpl_planner/common/planner_base.h>): see map_planner.h */
#ifndef MPLX_SHIM_PLANNER_BASE_H
#define MPLX_SHIM_PLANNER_BASE_H
#include <mpl_planner/planner/map_planner.h>
#endif
