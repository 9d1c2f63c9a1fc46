#!/bin/bash
# the driver's own commands on the round's last commit: pytest in its -x -q form, smoke(), the default bench line
OUT=$PWD/gpurun_out/${1:-r06u}; mkdir -p $OUT
export MPLX_DEADLINE_S=120 TMPDIR=/tmp
(timeout 1200 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6) > $OUT/pytest_gpu_driver_form.txt 2>&1; tail -2 $OUT/pytest_gpu_driver_form.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
T0=$(date +%s); timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench wall $(( $(date +%s) - T0 )) s" | tee $OUT/bench_default.wall
python tools/closing_numbers.py $OUT 2>/dev/null | head -9 | cut -c1-300
