#!/bin/bash
# Round 5: soak of the fill-load repeat (>= 50 batches): both C4 batches planned again and again under the background fill load.
set -u
OUT=gpurun_out/${1:-r05i}; mkdir -p $OUT
export MPLX_DEADLINE_S=100
timeout 300 python tools/fill_load_probe.py 30 fill acc > $OUT/fill_acc.json 2> $OUT/fill_acc.err; cut -c1-300 $OUT/fill_acc.json; python -c "import json; d=json.load(open('$OUT/fill_acc.json')); print('ACC', d['batches'], 'batches, fill rounds', d['fill_rounds'], 'mismatching queries', d['mismatching_queries'])"
timeout 200 python tools/fill_load_probe.py 60 fill jrk > $OUT/fill_jrk.json 2> $OUT/fill_jrk.err; python -c "import json; d=json.load(open('$OUT/fill_jrk.json')); print('JRK', d['batches'], 'batches, fill rounds', d['fill_rounds'], 'mismatching queries', d['mismatching_queries'])"
