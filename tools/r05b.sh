#!/bin/bash
# Round 5, GPU call B: the JRK batch's repeatability (helpers off = reference; plain state loads = round 4's kernel; the fix), then the
# whole -m gpu suite with durations under the launch guard.
set -u
OUT=gpurun_out/${1:-r05b}; mkdir -p $OUT
export MPLX_DEADLINE_S=100
timeout 300 python tools/r05_jrk_batch.py 60 768 0:0:2 -1:128:3 -1:0:4 > $OUT/jrk_batch.json 2> $OUT/jrk_batch.err; tail -c 2500 $OUT/jrk_batch.err
(timeout 1000 python -u -m pytest tests -m gpu -x -q --durations=40 2>&1 | tail -80) > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt
