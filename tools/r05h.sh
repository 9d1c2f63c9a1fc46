#!/bin/bash
set -u
OUT=gpurun_out/${1:-r05h}; mkdir -p $OUT
export MPLX_DEADLINE_S=100
V=$PWD/build_tmp/libmplx_rowpairs.so
MPLX_LIB=$V timeout 200 python tools/r05_ab.py 4 0 > $OUT/ab_pairs.json 2> $OUT/ab_pairs.err; grep -v amdgpu $OUT/ab_pairs.err
timeout 200 python tools/r05_ab.py 4 0 > $OUT/ab_prod.json 2> $OUT/ab_prod.err; grep -v amdgpu $OUT/ab_prod.err
MPLX_LIB=$V timeout 200 python tools/r05_ab.py 4 0 > $OUT/ab_pairs_2.json 2> $OUT/ab_pairs_2.err; grep -v amdgpu $OUT/ab_pairs_2.err
timeout 200 python tools/r05_ab.py 4 0 > $OUT/ab_prod_2.json 2> $OUT/ab_prod_2.err; grep -v amdgpu $OUT/ab_prod_2.err
MPLX_LIB=$V timeout 300 python tools/fill_load_probe.py 10 fill acc > $OUT/fill_pairs.json 2> $OUT/fill_pairs.err; cut -c1-700 $OUT/fill_pairs.json
(MPLX_LIB=$V timeout 200 python -u -m pytest tests/test_gpu_parity.py "tests/test_gpu_scale.py::test_helpers_leave_every_result_unchanged" "tests/test_gpu_scale.py::test_c2_full_query_acc_256" -m gpu -x -q 2>&1 | tail -3) > $OUT/pytest_pairs.txt; cat $OUT/pytest_pairs.txt
