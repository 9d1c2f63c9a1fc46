#!/bin/bash
# Round 5, GPU call E: the JRK batch after the counter-read fix (helpers off = reference, then 6 runs with helpers), then the whole
# -m gpu suite with durations under the launch guard.
set -u
OUT=gpurun_out/${1:-r05e}; mkdir -p $OUT
export MPLX_DEADLINE_S=100
timeout 300 python tools/r05_jrk_batch.py 60 768 0:0:1 -1:0:6 > $OUT/jrk_batch.json 2> $OUT/jrk_batch.err; cut -c1-400 $OUT/jrk_batch.err
(timeout 1100 python -u -m pytest tests -m gpu -x -q --durations=30 2>&1 | tail -60) > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt
