#!/bin/bash
# same-box A/B: what the epoch-tagged table and the pool recycling are worth on the blocking step and on the streamed leg
set -u
TAG=${1:-r06g}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp MPLX_DEADLINE_S=100
(timeout 1500 python -u -m pytest tests/test_pool_recycling.py tests/test_lpa.py tests/test_stream.py tests/test_gpu_parity.py tests/test_poly_map.py tests/test_guard.py tests/test_gpu_scale.py -m gpu -x -q --durations=10 2>&1 | tail -30) > $OUT/pytest.txt 2>&1; tail -16 $OUT/pytest.txt
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --stream 8 --extras 0 > $OUT/bench_$name.json 2> $OUT/bench_$name.err
python - <<PY
import json
d=json.load(open("$OUT/bench_$name.json")); s=d.get("stream",{})
print("$name: value", round(d["value"]/1e6,2), "ms", round(d["ms_per_step"],1), "pool_full", d["plan_status_counts"]["pool_full"], "| stream", round(s.get("value",0)/1e6,1), "steady", round(s.get("steady_state_ms_per_batch") or 0,1), "mism", s.get("parity",{}).get("mismatches_vs_blocking_step"), s.get("error"))
PY
}
run recycle_epoch A=1
run norecycle_epoch MPLX_BENCH_NO_RECYCLE=1
run norecycle_clear MPLX_BENCH_NO_RECYCLE=1 MPLX_TABLE_CLEAR=1
run recycle_clear MPLX_TABLE_CLEAR=1
run recycle_epoch2 A=1
