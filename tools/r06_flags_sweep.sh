#!/bin/bash
# compiler-option A/B of the 27-input ACC kernels (variant libraries from tools/build_kernel_variant.sh <name> "-DMPLX_ONLY_ACC <options>" "help spec")
# and a C2 knob sweep (wider OPEN buckets, more helpers per leader) on one box
set -u
TAG=${1:-r06ab}
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
for v in f_base f_o2 f_maxilp f_trackers f_base; do one $v "tail bulk c2" MPLX_LIB=$PWD/build_tmp/libmplx_$v.so; done
one c2_base c2 A=1
for w in 120 160 240 320; do one c2_bw$w c2 AB_BUCKET_WIDTH=$w; done
for h in "6,64" "8,64"; do one c2_h${h/,/_} c2 AB_HELPERS=$h; done
