#!/bin/bash
# A/B variants of the moving-obstacle kernels only (mplx_poly_launch.hip), linked with the product build's other objects.
# usage: tools/build_poly_variant.sh <name> "<-D flags>"      (use with MPLX_LIB=build_tmp/libmplx_<name>.so)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
N=$1; DEF=$2
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function $DEF"
S=$ROOT/mpl_ros_amd/csrc
O=$ROOT/build_tmp
mkdir -p $O
/opt/rocm/bin/hipcc $F -c -o $O/pv_$N.o $S/mplx_poly_launch.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libmplx_$N.so $S/mplx_api.o $S/mplx_spec_launch.o $S/mplx_help_launch.o $S/mplx_yaw_launch.o $S/mplx_lpa_launch.o $O/pv_$N.o $S/mplx_host.o
rm -f $O/pv_$N.o
ls -la $O/libmplx_$N.so
