#!/bin/bash
# Round 5, GPU call C: bisecting the JRK batch's run-to-run differences with the diagnostic switches + the record audit; the launch guard tests.
set -u
OUT=gpurun_out/${1:-r05c}; mkdir -p $OUT
export MPLX_DEADLINE_S=100
timeout 400 python tools/r05_jrk_batch.py 60 768 0:0:1 -1:0:4 -1:256:3 -1:512:3 -1:768:2 -1:128:2 > $OUT/jrk_batch.json 2> $OUT/jrk_batch.err; tail -c 6000 $OUT/jrk_batch.err
(timeout 300 python -u -m pytest tests/test_guard.py -m gpu -x -q 2>&1 | tail -30) > $OUT/pytest_guard.txt; cat $OUT/pytest_guard.txt
