#!/bin/bash
# A/B of variant libraries of the 27-input ACC kernels -- the merged refill (pull_fine_run): m_off / m_<target> (tools/build_kernel_variant.sh m_<t>
# "-DMPLX_ONLY_ACC -DMPLX_MERGE_TARGET=<t>" "help spec"), compiler options: f_<name> (... f_o2 "-DMPLX_ONLY_ACC -O2" "help spec") -- crossed with the coarse
# bucket width (MPLX_BUCKET_FACTOR x w dt), on one box.  VARIANTS / FACTORS / MODES / BLOCKS select; C2 knob runs: VARIANTS= with AB_BUCKET_WIDTH in the environment
set -u
TAG=${1:-r06ac}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp MPLX_DEADLINE_S=100
one() { name=$1; modes=$2; shift 2; env "$@" timeout 300 python tools/ab.py 2 $modes > $OUT/ab_$name.json 2> $OUT/ab_$name.err
python - <<PY
import json
try:
    d=json.load(open("$OUT/ab_$name.json")); print("$name", {m:(d[m]["min_ms"], d[m]["mean_ms"], d[m]["digests"][0][:6]) for m in d if m!="lib"})
except Exception as e: print("$name failed", e)
PY
}
for v in ${VARIANTS:-m_off m_64 m_128 m_256}; do
  for f in ${FACTORS:-8 4 2}; do one ${v}_f$f "${MODES:-tail bulk c2}" MPLX_LIB=$PWD/build_tmp/libmplx_$v.so MPLX_BUCKET_FACTOR=$f; done
done
for spec in ${BLOCKS:-m_off:8 m_128:8 m_128:3}; do v=${spec%%:*}; f=${spec##*:}; one ${v}_f${f}_block block MPLX_LIB=$PWD/build_tmp/libmplx_$v.so MPLX_BUCKET_FACTOR=$f; done
