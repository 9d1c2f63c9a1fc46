#!/bin/bash
# Round 5, GPU call A: the launch guard, then the batch the round-4 driver run stalled in, then the whole -m gpu suite with durations.
set -u
OUT=gpurun_out/${1:-r05a}; mkdir -p $OUT
export MPLX_DEADLINE_S=100
(timeout 240 python -m pytest tests/test_guard.py -m gpu -x -q 2>&1 | tail -15) > $OUT/pytest_guard.txt; cat $OUT/pytest_guard.txt
timeout 400 python tools/r05_jrk_batch.py 3 90 768 -1 > $OUT/jrk_batch.json 2> $OUT/jrk_batch.err; tail -c 3000 $OUT/jrk_batch.err
(timeout 1100 python -m pytest tests -m gpu -x -q --durations=40 2>&1 | tail -70) > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt
