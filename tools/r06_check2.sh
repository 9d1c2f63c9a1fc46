#!/bin/bash
# GPU check of round 6's state-table / recycling / re-root changes: the new tests, the LPA* and stream tests, then the C4 batch with
# recycling at full pool size (per-query timing dump -> how small the pools may get)
set -u
TAG=${1:-r06e}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp MPLX_DEADLINE_S=100
(timeout 1200 python -u -m pytest tests/test_pool_recycling.py tests/test_lpa.py tests/test_stream.py tests/test_gpu_parity.py tests/test_poly_map.py tests/test_guard.py -m gpu -x -q --durations=10 2>&1 | tail -30) > $OUT/pytest.txt 2>&1; tail -14 $OUT/pytest.txt
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 8 --stream 4 --extras 0 --dump-queries $OUT/queries.json > $OUT/bench_recycle.json 2> $OUT/bench_recycle.err
python - <<PY
import json
d=json.load(open("$OUT/bench_recycle.json"))
print("value", round(d["value"]/1e6,2), "ms", round(d["ms_per_step"],1), "status", d["plan_status_counts"], "parity", d.get("parity_sample"), "stream", round(d.get("stream",{}).get("value",0)/1e6,1), d.get("stream",{}).get("parity",{}).get("mismatches_vs_blocking_step"), d.get("stream",{}).get("error"))
print("spec", d.get("speculation"))
q=json.load(open("$OUT/queries.json"))
import numpy as np
tb,te,nn=np.array(q["t_begin"]),np.array(q["t_end"]),np.array(q["n_nodes"])
ev=sorted([(t,+n) for t,n in zip(tb,nn)]+[(t,-n) for t,n in zip(te,nn)])
cur=peak=0
for t,dn in ev:
    cur+=dn; peak=max(peak,cur)
print("states: total", int(nn.sum()), "peak concurrently held (final sizes)", int(peak), "ratio", round(peak/nn.sum(),3), "biggest", int(nn.max()))
PY
