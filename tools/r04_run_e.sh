#!/bin/bash
# round-4 GPU run E: new LPA* rule on the device, the reference's robot code on the device back-end, repeatability at bench
# size, helper-limit sweep of the streamed leg.  Every step has its own timeout; the sum stays under 12 minutes.
set -u
OUT=gpurun_out/r04e; mkdir -p $OUT
(timeout 420 python -m pytest tests/test_lpa.py "tests/test_cpp_shim.py::test_reference_robot_team_plans_through_the_backend_unchanged" "tests/test_gpu_scale.py::test_c4_acc_batch_repeats_blocking_and_streamed" -m gpu -x -q -s --timeout 200 2>&1 | tail -25) > $OUT/pytest.txt 2>&1
tail -12 $OUT/pytest.txt
run() { name=$1; shift; timeout 110 python bench.py --cpu-seconds 0 --steps 1 --warmup 0 --stream 8 "$@" > $OUT/$name.json 2> $OUT/$name.err; python -c "
import json
d=json.load(open('$OUT/$name.json'))
s=d.get('stream') or {}
print('$name: blocking', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],1), 'ms | stream', round(s.get('value',0)/1e6,2), 'M/s', s.get('ms_per_batch'), s.get('steady_state_ms_per_batch'), s.get('batch_latency_ms'), json.dumps(s.get('parity'))[:900])
" 2>&1 | tail -2; }
run lim64 --stream-helper-limit 64
run lim96 --stream-helper-limit 96
run lim64_res32 --stream-helper-limit 64 --stream-reserved 32
