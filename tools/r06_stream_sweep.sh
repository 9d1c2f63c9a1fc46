#!/bin/bash
# streamed leg of the bench line against the helper limit (workgroups of a launch that stay on as helpers once its queue is empty) and the reserved share, one box
set -u
TAG=${1:-r06ak}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp MPLX_DEADLINE_S=100
for spec in 32:0 16:0 24:0 48:0 64:0 32:16 32:0; do L=${spec%%:*}; R=${spec##*:}
  timeout 200 python bench.py --steps 1 --warmup 1 --cpu-seconds 0 --stream 10 --extras 0 --stream-helper-limit $L --stream-reserved $R > $OUT/stream_l${L}_r$R.json 2> $OUT/stream_l${L}_r$R.err
  python -c "
import json
try:
    d=json.load(open('$OUT/stream_l${L}_r$R.json')); s=d['stream']; print('limit $L reserved $R: whole', round(s['value']/1e6,1), 'M/s, steady', round(s['steady_state_ms_per_batch'],1), 'ms/batch =', round(d['expansions_per_step']/s['steady_state_ms_per_batch']/1e3,1), 'M/s, mismatches', s['parity']['mismatches_vs_blocking_step'])
except Exception as e: print('limit $L reserved $R failed', e)"
done
