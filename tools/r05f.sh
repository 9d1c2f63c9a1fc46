#!/bin/bash
# Round 5, GPU call F: fill-load repeat tests (ACC, JRK; subprocesses), then the A/B of the round-4 safeguards on the blocking C4-ACC step.
set -u
OUT=gpurun_out/${1:-r05f}; mkdir -p $OUT
export MPLX_DEADLINE_S=100
(timeout 500 python -u -m pytest tests/test_zz_jitter.py -m gpu -q 2>&1 | tail -25) > $OUT/pytest_jitter.txt; cat $OUT/pytest_jitter.txt
timeout 400 python tools/r05_ab.py 3 > $OUT/ab.json 2> $OUT/ab.err; cat $OUT/ab.err | grep -v amdgpu.ids
