#!/bin/bash
# OPEN bucket width (MPLX_BUCKET_FACTOR x w dt; product: 8) on one box: C4-ACC block / bulk / C2 (tools/ab.py), the C4-JRK batch, the C3 query
set -u
TAG=${1:-r06m}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp MPLX_DEADLINE_S=100
for f in 8 4 3 2 1 8; do
  export MPLX_BUCKET_FACTOR=$f
  timeout 300 python tools/ab.py 2 block bulk c2 > $OUT/ab_f$f.json 2> $OUT/ab_f$f.err
  timeout 120 python bench.py --lattice jrk --steps 3 --warmup 1 --cpu-seconds 0 --stream 0 > $OUT/jrk_f$f.json 2> $OUT/jrk_f$f.err
  timeout 120 python bench.py --single --lattice jrk --steps 1 --warmup 1 --warmup-cap 20000 --cpu-seconds 0 > $OUT/c3_f$f.json 2> $OUT/c3_f$f.err
  python - <<PY
import json
def g(p, k="ms_per_step"):
    try: return round(json.load(open(p))[k], 1)
    except Exception as e: return "failed"
try:
    d = json.load(open("$OUT/ab_f$f.json")); ab = {m: (d[m]["min_ms"], d[m]["digests"][0][:6]) for m in d if m != "lib"}
except Exception as e: ab = "failed"
print("factor $f", ab, "jrk", g("$OUT/jrk_f$f.json"), "c3", g("$OUT/c3_f$f.json"))
PY
done
