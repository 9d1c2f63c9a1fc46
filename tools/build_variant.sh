#!/bin/bash
# Builds a diagnostic variant of libmplx.so into build_tmp/ (git-ignored; travels with gpurun snapshots).
# usage: tools/build_variant.sh timers|helpdbg|rowpairs|earlytomb|fast|claimwait1n|diag  -> build_tmp/libmplx_<variant>.so   (use with MPLX_LIB=...)
#   rowpairs / earlytomb / fast (= both): the two A/B switches round 4 hoped would give back the cost of its safeguards -- measured in
#   round 5: rowpairs 15 % slower, earlytomb no gain; kept for re-measurement.  diag: the MPLX_X_FLAGS switches.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
V=${1:-timers}
case $V in
  timers) DEF="-DMPLX_LOOKUP_TIMERS -DMPLX_PHASE_TIMERS=1" ;;
  helpdbg) DEF=-DMPLX_HELP_DEBUG ;;
  rowpairs) DEF=-DMPLX_X_ROW_PAIRS=1 ;;
  earlytomb) DEF=-DMPLX_X_EARLY_TOMB=1 ;;
  fast) DEF="-DMPLX_X_ROW_PAIRS=1 -DMPLX_X_EARLY_TOMB=1" ;;
  claimwait1n) DEF=-DMPLX_X_CLAIM_WAIT_1N=1 ;;
  diag) DEF=-DMPLX_DIAG_FLAGS=1 ;;               # the MPLX_X_FLAGS switches of mplx_kernels.h (round 5's r05_jrk_batch.py, r05_ab.py)   # rule R3 for the one-node kernels (VEL / SNP states, LPA*, lattices > 128): test under tools/r04_jitter_probe.py-style load
  *) echo "unknown variant $V"; exit 2 ;;
esac
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function $DEF"
S=$ROOT/mpl_ros_amd/csrc
O=$ROOT/build_tmp
mkdir -p $O
/opt/rocm/bin/hipcc $F -c -o $O/v_api.o $S/mplx_api.hip &
/opt/rocm/bin/hipcc $F -c -o $O/v_spec.o $S/mplx_spec_launch.hip &
/opt/rocm/bin/hipcc $F -c -o $O/v_help.o $S/mplx_help_launch.hip &
/opt/rocm/bin/hipcc $F -c -o $O/v_yaw.o $S/mplx_yaw_launch.hip &
/opt/rocm/bin/hipcc $F -c -o $O/v_lpa.o $S/mplx_lpa_launch.hip &
/opt/rocm/bin/hipcc $F -c -o $O/v_poly.o $S/mplx_poly_launch.hip &
/opt/rocm/bin/hipcc $F -c -o $O/v_filter.o $S/mplx_filter_launch.hip &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libmplx_$V.so $O/v_api.o $O/v_spec.o $O/v_help.o $O/v_yaw.o $O/v_lpa.o $O/v_poly.o $O/v_filter.o $S/mplx_host.o
rm -f $O/v_*.o
ls -la $O/libmplx_$V.so
