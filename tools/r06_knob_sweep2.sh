#!/bin/bash
# host-side knobs on the shipped library, one box: the C5 tick and the C3 query against the coarse OPEN bucket width (MPLX_BUCKET_FACTOR x w dt)
set -u
TAG=${1:-r06aj}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp MPLX_DEADLINE_S=100
g() { python -c "
import json,sys
try: d=json.load(open('$1')); print('$2', round(d['ms_per_step'],1), 'ms', (d.get('parity_sample') or {}).get('mismatches'))
except Exception as e: print('$2 failed', e)"; }
for f in 8 3 16 32 64 8; do
  MPLX_BUCKET_FACTOR=$f timeout 100 python bench.py --config c5 --steps 4 --warmup 1 --cpu-seconds 0 > $OUT/c5_f$f.json 2> $OUT/c5_f$f.err; g $OUT/c5_f$f.json "c5 factor $f"
done
for f in 1 0.5 0.25; do
  MPLX_BUCKET_FACTOR=$f timeout 120 python bench.py --single --lattice jrk --steps 1 --warmup 1 --warmup-cap 20000 --cpu-seconds 0 > $OUT/c3_f$f.json 2> $OUT/c3_f$f.err; g $OUT/c3_f$f.json "c3 factor $f"
done
