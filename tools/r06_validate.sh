#!/bin/bash
# validation of a new library on one box: the -m gpu suite in the driver's form, smoke, the default bench line, then the C3 query and the capped C4-JRK
# batch against the coarse bucket width (MPLX_BUCKET_FACTOR), then variant libraries of the ACC kernel units named in VARIANTS (tools/build_kernel_variant.sh)
set -u
TAG=${1:-r06af}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export MPLX_DEADLINE_S=120 TMPDIR=/tmp
(timeout 1200 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6) > $OUT/pytest_gpu_driver_form.txt 2>&1; tail -2 $OUT/pytest_gpu_driver_form.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
T0=$(date +%s); timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench wall $(( $(date +%s) - T0 )) s" | tee $OUT/bench_default.wall
python tools/closing_numbers.py $OUT 2>/dev/null | head -9 | cut -c1-300
for f in 1 2 3; do
  export MPLX_BUCKET_FACTOR=$f
  timeout 120 python bench.py --lattice jrk --steps 3 --warmup 1 --cpu-seconds 0 --stream 0 > $OUT/jrk_f$f.json 2> $OUT/jrk_f$f.err
  timeout 120 python bench.py --single --lattice jrk --steps 1 --warmup 1 --warmup-cap 20000 --cpu-seconds 0 > $OUT/c3_f$f.json 2> $OUT/c3_f$f.err
  python - <<PY
import json
def g(p, k="ms_per_step"):
    try: return round(json.load(open(p))[k], 1)
    except Exception as e: return "failed"
print("jrk factor $f: capped C4-JRK batch", g("$OUT/jrk_f$f.json"), "ms, C3", g("$OUT/c3_f$f.json"), "ms")
PY
done
unset MPLX_BUCKET_FACTOR
one() { name=$1; modes=$2; shift 2; env "$@" timeout 300 python tools/ab.py 2 $modes > $OUT/ab_$name.json 2> $OUT/ab_$name.err
python - <<PY
import json
try:
    d=json.load(open("$OUT/ab_$name.json")); print("$name", {m:(d[m]["min_ms"], d[m]["mean_ms"], d[m]["digests"][0][:6]) for m in d if m!="lib"})
except Exception as e: print("$name failed", e)
PY
}
one product "tail bulk c2 block" A=1
for v in ${VARIANTS:-}; do one $v "tail bulk c2" MPLX_LIB=$PWD/build_tmp/libmplx_$v.so MPLX_BUCKET_FACTOR=3; done  # (r06af: VARIANTS="m_512o2 o_os o_o1 o_o2nu")
