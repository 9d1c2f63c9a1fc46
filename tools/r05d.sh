#!/bin/bash
set -u
OUT=gpurun_out/${1:-r05d}; mkdir -p $OUT
export MPLX_DEADLINE_S=100
timeout 400 python tools/r05_jrk_batch.py 60 768 0:0:1 -1:768:3 -1:0:2 > $OUT/jrk_batch.json 2> $OUT/jrk_batch.err; tail -c 1500 $OUT/jrk_batch.err
