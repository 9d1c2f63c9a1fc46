#!/bin/bash
# First GPU call of the next round (about 9 minutes; every step has its own timeout, the sum stays below 11 minutes):
#   A. the product build: the driver's bench line, the blocking batch under a background fill load (the race of DESIGN.md 3.9 only
#      shows there), rocprofv3 trace of the final kernel (round 4 ended without one);
#   B. the A/B variant that should give the 7 % of the round-4 fix back (tools/build_variant.sh fast -> build_tmp/libmplx_fast.so,
#      built on the CPU box BEFORE this call): short bench, the fill-load probe (twice as long: it must stay at 0), a longer stream.
# Decide from B: blocking ms per step (product ~2310, before the fix 2151), mismatches (must be 0 everywhere).
set -u
TAG=${1:-r05a}
OUT=gpurun_out/$TAG; mkdir -p $OUT
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); s = d.get("stream") or {}; p = s.get("parity") or {}
    print(sys.argv[1], "blocking", round(d["ms_per_step"], 1), "ms", round(d["value"] / 1e6, 2), "M/s | parity", d.get("parity_sample"),
          "| stream", round(s.get("value", 0) / 1e6, 1), "M/s, mismatches", p.get("mismatches_vs_blocking_step"), "of", p.get("batches_checked"), s.get("error"))
except Exception as e:
    print(sys.argv[1], "failed:", e)
PY
}
probe() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(sys.argv[1], "fill-load probe:", d["mismatching_queries"], "differing queries in", d["batches"], "batches", d["detail"][:4])
except Exception as e:
    print(sys.argv[1], "failed:", e)
PY
}
# ---- A. product build
MPLX_BENCH_STREAM_STALL_S=40 timeout 170 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; line $OUT/bench_default.json
timeout 70 python tools/r04_jitter_probe.py 10 fill > $OUT/jitter_product.json 2> $OUT/jitter_product.err; probe $OUT/jitter_product.json
( cd /tmp && TMPDIR=/tmp timeout 100 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/trace -o bench -- python $OLDPWD/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --stream 0 > $OLDPWD/$OUT/trace.log 2>&1 )
python profiles/summarize_rocprof.py $OUT/trace > $OUT/summary_trace.txt 2>&1; grep -m 3 astar_spec $OUT/summary_trace.txt | cut -c1-200; find $OUT/trace -name "*.db" -delete
# ---- B. the variant
if [ -f build_tmp/libmplx_fast.so ]; then
  export MPLX_LIB=$PWD/build_tmp/libmplx_fast.so
  MPLX_BENCH_STREAM_STALL_S=40 timeout 110 python bench.py --gpus 1 --steps 8 --warmup 2 --stream 24 > $OUT/bench_fast.json 2> $OUT/bench_fast.err; line $OUT/bench_fast.json
  timeout 100 python tools/r04_jitter_probe.py 20 fill > $OUT/jitter_fast.json 2> $OUT/jitter_fast.err; probe $OUT/jitter_fast.json
else
  echo "build_tmp/libmplx_fast.so missing: run tools/build_variant.sh fast first"
fi
# ---- C. rule R3 for the one-node kernels (tools/build_variant.sh claimwait1n): the parity tests that run them must stay green
if [ -f build_tmp/libmplx_claimwait1n.so ]; then
  (MPLX_LIB=$PWD/build_tmp/libmplx_claimwait1n.so timeout 120 python -m pytest tests/test_lpa.py tests/test_gpu_parity.py tests/test_poly_map.py -m gpu -x -q 2>&1 | tail -3) > $OUT/pytest_claimwait1n.txt; cat $OUT/pytest_claimwait1n.txt
fi
