#!/bin/bash
set -u
OUT=gpurun_out/${1:-r05g}; mkdir -p $OUT
export MPLX_DEADLINE_S=100
MPLX_LIB=$PWD/build_tmp/libmplx_r05z.so timeout 200 python tools/r05_ab.py 4 0 > $OUT/ab_r05z.json 2> $OUT/ab_r05z.err; grep -v amdgpu $OUT/ab_r05z.err
timeout 200 python tools/r05_ab.py 4 0 > $OUT/ab_new.json 2> $OUT/ab_new.err; grep -v amdgpu $OUT/ab_new.err
MPLX_LIB=$PWD/build_tmp/libmplx_r05z.so timeout 200 python tools/r05_ab.py 4 0 > $OUT/ab_r05z_2.json 2> $OUT/ab_r05z_2.err; grep -v amdgpu $OUT/ab_r05z_2.err
timeout 200 python tools/r05_ab.py 4 0 > $OUT/ab_new_2.json 2> $OUT/ab_new_2.err; grep -v amdgpu $OUT/ab_new_2.err
(timeout 200 python -u -m pytest tests/test_guard.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3) > $OUT/pytest_subset.txt; cat $OUT/pytest_subset.txt
