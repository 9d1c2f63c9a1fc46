#!/bin/bash
# A/B variants of the helper-assisted speculative kernel only: recompiles mplx_help_launch.hip with extra -D flags and links
# it with the product build's other objects (make -C mpl_ros_amd/csrc first) into build_tmp/libmplx_<name>.so.
# usage: tools/build_kernel_variant.sh <name> "<-D flags>"      (use with MPLX_LIB=build_tmp/libmplx_<name>.so)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
N=$1; DEF=$2
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function $DEF"
S=$ROOT/mpl_ros_amd/csrc
O=$ROOT/build_tmp
mkdir -p $O
/opt/rocm/bin/hipcc $F -c -o $O/kv_$N.o $S/mplx_help_launch.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libmplx_$N.so $S/mplx_api.o $S/mplx_spec_launch.o $O/kv_$N.o $S/mplx_yaw_launch.o $S/mplx_lpa_launch.o $S/mplx_poly_launch.o $S/mplx_host.o
rm -f $O/kv_$N.o
ls -la $O/libmplx_$N.so
