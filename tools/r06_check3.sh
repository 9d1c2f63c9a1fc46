#!/bin/bash
set -u
TAG=${1:-r06f}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp MPLX_DEADLINE_S=100
(timeout 1500 python -u -m pytest tests/test_pool_recycling.py tests/test_lpa.py tests/test_stream.py tests/test_gpu_parity.py tests/test_poly_map.py tests/test_guard.py tests/test_gpu_scale.py -m gpu -x -q --durations=10 2>&1 | tail -30) > $OUT/pytest.txt 2>&1; tail -16 $OUT/pytest.txt
for depth in 2 3; do
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 8 --stream 8 --stream-depth $depth --extras 0 > $OUT/bench_d$depth.json 2> $OUT/bench_d$depth.err
python - <<PY
import json
d=json.load(open("$OUT/bench_d$depth.json"))
s=d.get("stream",{})
print("depth $depth: value", round(d["value"]/1e6,2), "ms", round(d["ms_per_step"],1), "status", d["plan_status_counts"], "parity", d.get("parity_sample"), "pools", d["config"].get("pools"))
print("   stream", round(s.get("value",0)/1e6,1), "steady", s.get("steady_state_ms_per_batch"), "mism", s.get("parity",{}).get("mismatches_vs_blocking_step"), s.get("error"))
PY
done
nvidia-smi >/dev/null 2>&1; rocm-smi --showmeminfo vram 2>/dev/null | tail -4
