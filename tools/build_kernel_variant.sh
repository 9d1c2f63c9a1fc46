#!/bin/bash
# A/B variants of the speculative kernels only: recompiles the named launch translation units with extra -D flags and links them
# with the product build's other objects (make -C mpl_ros_amd/csrc first) into build_tmp/libmplx_<name>.so.
# usage: tools/build_kernel_variant.sh <name> "<-D flags>" [units: help spec filter yaw lpa poly api, default "help"]
#        (use with MPLX_LIB=build_tmp/libmplx_<name>.so; -DMPLX_ONLY_ACC builds the 27-input ACC kernel of a unit alone: ~15 s)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
N=$1; DEF=$2; UNITS=${3:-help}
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function $DEF"
S=$ROOT/mpl_ros_amd/csrc
O=$ROOT/build_tmp
mkdir -p $O
OBJS=""
for u in api spec help yaw lpa poly filter; do
  src=mplx_${u}_launch; [ $u = api ] && src=mplx_api
  if [[ " $UNITS " == *" $u "* ]]; then
    OPT=""; case $u in spec|help|yaw|filter) OPT="-O2";; esac  # (the Makefile's FLAGS_SPEC: the speculative kernel's units at -O2; a -O of $DEF still wins)
    /opt/rocm/bin/hipcc ${F/ $DEF/} $OPT $DEF -c -o $O/kv_${N}_$u.o $S/$src.hip &
    OBJS="$OBJS $O/kv_${N}_$u.o"
  else
    OBJS="$OBJS $S/$src.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libmplx_$N.so $OBJS $S/mplx_poly_lpa.o $S/mplx_host.o
rm -f $O/kv_${N}_*.o
ls -la $O/libmplx_$N.so
