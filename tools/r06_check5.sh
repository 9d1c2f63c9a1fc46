#!/bin/bash
# same-box A/B of the state table's size (slots per pool state) on the recycled C4-ACC batch
set -u
TAG=${1:-r06i}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp MPLX_DEADLINE_S=100
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --stream 6 --extras 0 > $OUT/bench_$name.json 2> $OUT/bench_$name.err
python - <<PY
import json
d=json.load(open("$OUT/bench_$name.json")); s=d.get("stream",{})
print("$name: value", round(d["value"]/1e6,2), "ms", round(d["ms_per_step"],1), "internal", d["plan_status_counts"]["internal"], "| stream", round(s.get("value",0)/1e6,1), "steady", round(s.get("steady_state_ms_per_batch") or 0,1), "mism", s.get("parity",{}).get("mismatches_vs_blocking_step"), s.get("error"))
PY
}
run f8 A=1
run f4 MPLX_TABLE_FACTOR=4
run f2 MPLX_TABLE_FACTOR=2
run f8b A=1
run f16 MPLX_TABLE_FACTOR=16
