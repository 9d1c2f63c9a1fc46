#!/bin/bash
# round-4 GPU run F: split batches over more lanes (every step has its own timeout)
set -u
OUT=gpurun_out/r04f; mkdir -p $OUT
run() { name=$1; shift; timeout 120 python bench.py --cpu-seconds 0 --steps 1 --warmup 0 --stream 10 "$@" > $OUT/$name.json 2> $OUT/$name.err; python -c "
import json
d=json.load(open('$OUT/$name.json'))
s=d.get('stream') or {}
print('$name: stream', round(s.get('value',0)/1e6,2), 'M/s', s.get('ms_per_batch'), s.get('steady_state_ms_per_batch'), s.get('batch_latency_ms'), json.dumps(s.get('parity'))[:700])
" 2>&1 | tail -2; tail -n 3 $OUT/$name.err | grep -v amdgpu.ids; }
run s2d4l16 --stream-split 2 --stream-depth 4 --stream-helper-limit 16
run s2d4l32 --stream-split 2 --stream-depth 4 --stream-helper-limit 32
run s4d6l8 --stream-split 4 --stream-depth 6 --stream-helper-limit 8
