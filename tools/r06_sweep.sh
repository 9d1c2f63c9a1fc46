#!/bin/bash
# run-time knob sweep on one box (product library): OPEN bucket width, helpers per leader x reserved share
set -u
TAG=${1:-r06k}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp MPLX_DEADLINE_S=100
one() { name=$1; shift; env "$@" timeout 300 python tools/ab.py 2 tail block > $OUT/ab_$name.json 2> $OUT/ab_$name.err
python - <<PY
import json
try:
    d=json.load(open("$OUT/ab_$name.json")); print("$name", {m:(d[m]["min_ms"], d[m]["mean_ms"], d[m]["digests"][0][:6]) for m in d if m!="lib"})
except Exception as e: print("$name failed", e)
PY
}
one base A=1
for w in 20 40 160 320; do one bw$w AB_BUCKET_WIDTH=$w; done
one base2 A=1
for h in "2,32" "3,48" "4,96" "4,128"; do one h${h/,/_} AB_HELPERS=$h; done
