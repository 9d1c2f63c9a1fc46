#!/bin/bash
# Round-4 closing evidence on one GPU box (every step has its own timeout; ~13 minutes in all):
#   1. the whole -m gpu suite   2. the default bench line (the driver's command)   3. C5 / LPA / C3 lines   4. rocprofv3 passes
set -u
TAG=${1:-r04z}
OUT=gpurun_out/$TAG; mkdir -p $OUT
(timeout 480 python -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -15) > $OUT/pytest_gpu.txt 2>&1
tail -4 $OUT/pytest_gpu.txt
timeout 260 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_default.json")); s=d.get("stream") or {}
    print("default: blocking", round(d["value"]/1e6,2), "M/s", round(d["ms_per_step"],1), "ms frac", d["roofline"]["frac"], "| stream", round(s.get("value",0)/1e6,2), "M/s", s.get("steady_state_ms_per_batch"), s.get("batch_latency_ms"), s.get("parity",{}).get("mismatches_vs_blocking_step"), "| parity", d.get("parity_sample"), "| cpu", d.get("cpu_baseline",{}).get("value"))
except Exception as e:
    print("bench default failed", e)
PY
timeout 100 python bench.py --config c5 --steps 3 --warmup 1 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; head -c 400 $OUT/bench_c5.json; echo
timeout 60 python bench.py --config lpa --steps 2 --warmup 1 > $OUT/bench_lpa.json 2> $OUT/bench_lpa.err; head -c 300 $OUT/bench_lpa.json; echo
timeout 120 python bench.py --single --lattice jrk --steps 1 --warmup 1 --cpu-seconds 0 > $OUT/bench_c3.json 2> $OUT/bench_c3.err; head -c 300 $OUT/bench_c3.json; echo
tools/profile_r04.sh $TAG/prof
timeout 330 python tools/full_parity_c4.py $OUT/full_parity_c4.json > $OUT/full_parity.log 2>&1; tail -n 1 $OUT/full_parity.log | cut -c1-600
