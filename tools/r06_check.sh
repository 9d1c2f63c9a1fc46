#!/bin/bash
# quick GPU check of the working tree: selected -m gpu tests + the driver's bench command (with the extra configurations)
set -u
TAG=${1:-r06d}; shift
TESTS=${*:-"tests/test_guard.py tests/test_cpp_shim.py tests/test_lpa.py tests/test_gpu_parity.py"}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
(timeout 900 python -u -m pytest $TESTS -m gpu -x -q --durations=8 2>&1 | tail -25) > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
T0=$(date +%s); timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench wall $(( $(date +%s) - T0 )) s"
grep "bench +" $OUT/bench_default.err | tail -30 | cut -c1-200
python - <<PY
import json
d=json.load(open("$OUT/bench_default.json"))
print("value", round(d["value"]/1e6,2), "ms", round(d["ms_per_step"],1), "frac", round(d["roofline"]["frac"],5), "spec", d.get("speculation"))
print("stream", round(d.get("stream",{}).get("value",0)/1e6,2), d.get("stream",{}).get("parity",{}).get("mismatches_vs_blocking_step"), "parity", d.get("parity_sample"))
for k in ("c2","c3","c5","lpa"):
    e=d.get(k,{})
    print(k, {x: e.get(x) for x in ("value","ms_per_step","leg_seconds","error","vs_cpu_single_thread","parity_sample")}, "frac", e.get("roofline",{}).get("frac"), "cpu", e.get("cpu_baseline",{}).get("value"))
PY
