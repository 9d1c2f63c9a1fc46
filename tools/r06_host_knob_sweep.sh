#!/bin/bash
# host-side knobs on the shipped library, one box (nothing here changes device code):
#   width  : the C5 tick and the C3 query against the coarse OPEN bucket width (MPLX_BUCKET_FACTOR x w dt)
#   stream : the streamed leg of the bench line against the helper limit (workgroups of a launch that stay on as helpers once its queue
#            is empty) and the reserved share
# usage: tools/r06_host_knob_sweep.sh <tag> [width stream]
set -u
TAG=${1:-r06aj}; shift
STEPS=${*:-"width stream"}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp MPLX_DEADLINE_S=100
g() { python -c "
import json,sys
try: d=json.load(open('$1')); print('$2', round(d['ms_per_step'],1), 'ms', (d.get('parity_sample') or {}).get('mismatches'))
except Exception as e: print('$2 failed', e)"; }
for s in $STEPS; do
case $s in
width)
  for f in 8 3 16 32 64 8; do
    MPLX_BUCKET_FACTOR=$f timeout 100 python bench.py --config c5 --steps 4 --warmup 1 --cpu-seconds 0 > $OUT/c5_f$f.json 2> $OUT/c5_f$f.err; g $OUT/c5_f$f.json "c5 factor $f"
  done
  for f in 1 0.5 0.25; do
    MPLX_BUCKET_FACTOR=$f timeout 120 python bench.py --single --lattice jrk --steps 1 --warmup 1 --warmup-cap 20000 --cpu-seconds 0 > $OUT/c3_f$f.json 2> $OUT/c3_f$f.err; g $OUT/c3_f$f.json "c3 factor $f"
  done ;;
stream)
  for spec in 32:0 16:0 24:0 48:0 64:0 32:16 32:0; do L=${spec%%:*}; R=${spec##*:}
    timeout 200 python bench.py --steps 1 --warmup 1 --cpu-seconds 0 --stream 10 --extras 0 --stream-helper-limit $L --stream-reserved $R > $OUT/stream_l${L}_r$R.json 2> $OUT/stream_l${L}_r$R.err
    python -c "
import json
try:
    d=json.load(open('$OUT/stream_l${L}_r$R.json')); s=d['stream']; print('limit $L reserved $R: whole', round(s['value']/1e6,1), 'M/s, steady', round(s['steady_state_ms_per_batch'],1), 'ms/batch =', round(d['expansions_per_step']/s['steady_state_ms_per_batch']/1e3,1), 'M/s, mismatches', s['parity']['mismatches_vs_blocking_step'])
except Exception as e: print('limit $L reserved $R failed', e)"
  done ;;
esac
done
