#!/bin/bash
# Round-6 closing evidence on one GPU box, ONE library for every number (every step has its own timeout; ~25 minutes in all):
#   1. the whole -m gpu suite (per-test durations)          2. the default bench line (the driver's command, extras included)
#   3. rocprofv3 passes of the blocking C4-ACC launch (trace + FETCH / WRITE / SQ / L2 counters, each in its own pass), of the bulk
#      phase, kernel traces of the streamed leg, the C4-JRK batch, the C3 query and the C5 tick                 -> profiles/traffic.json
#   4. the other configurations as lines of their own (C5, LPA*, C3, C4-JRK capped at 20 000, C4-JRK at the survey's 2 M cap)
#   5. per-phase tables of the timers build (tail and bulk regime)
#   6. all 1024 queries of the C4 batches replayed on the CPU (ACC and JRK)
# usage: tools/r06_final.sh <tag> [steps: suite bench prof other phase replay]
set -u
TAG=${1:-r06x}; shift
STEPS=${*:-"suite bench prof other phase replay"}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export MPLX_DEADLINE_S=100 TMPDIR=/tmp
ROOT=$PWD
for s in $STEPS; do
case $s in
suite)
  (timeout 900 python -u -m pytest tests -m gpu -q --durations=25 2>&1 | tail -45) > $OUT/pytest_gpu.txt 2>&1
  tail -3 $OUT/pytest_gpu.txt ;;
bench)
  T0=$(date +%s)
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
  echo "bench wall $(( $(date +%s) - T0 )) s" | tee $OUT/bench_default.wall
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_default.json")); s=d.get("stream") or {}
    print("default: blocking", round(d["value"]/1e6,2), "M/s", round(d["ms_per_step"],1), "ms frac", round(d["roofline"]["frac"],5), "| stream", round(s.get("value",0)/1e6,2), "M/s", s.get("steady_state_ms_per_batch"), s.get("parity",{}).get("mismatches_vs_blocking_step"), "| parity", d.get("parity_sample"), "| cpu", d.get("cpu_baseline",{}).get("value"))
    print("speculation", d.get("speculation"))
    for k in ("c2","c3","c5","lpa","plpa"):
        e=d.get(k,{})
        print(k, {x: e.get(x) for x in ("value","ms_per_step","leg_seconds","error","vs_cpu_single_thread")}, "parity", (e.get("parity_sample") or {}).get("mismatches"), "frac", e.get("roofline",{}).get("frac"))
except Exception as e:
    print("bench default failed", e)
PY
  ;;
prof)
  BLOCK="python $ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --stream 0 --extras 0"
  BULK="python $ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --stream 0 --extras 0 --max-expand 20000 --helpers 0"
  mkdir -p $OUT/block $OUT/bulk $OUT/other
  cd /tmp
  pass() { d=$1; shift; what=$1; shift; timeout 170 rocprofv3 "$@" -d "$OUT/$d" -o bench -- $what > "$OUT/$d.log" 2>&1; tail -n 1 "$OUT/$d.log" | cut -c1-160; }
  pass block/trace "$BLOCK" --kernel-trace --stats
  pass block/pmc_fetch "$BLOCK" --pmc FETCH_SIZE
  pass block/pmc_write "$BLOCK" --pmc WRITE_SIZE
  pass block/pmc_sq1 "$BLOCK" --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES
  pass block/pmc_sq2 "$BLOCK" --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU
  pass block/pmc_l2 "$BLOCK" --pmc TCC_HIT_sum TCC_MISS_sum
  pass block/pmc_l2rd "$BLOCK" --pmc TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum
  pass bulk/trace "$BULK" --kernel-trace --stats
  pass bulk/pmc_sq1 "$BULK" --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES
  pass bulk/pmc_sq2 "$BULK" --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU
  pass bulk/pmc_l2 "$BULK" --pmc TCC_HIT_sum TCC_MISS_sum
  pass bulk/pmc_l2rd "$BULK" --pmc TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum
  pass other/trace_stream "python $ROOT/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --stream 6 --extras 0" --kernel-trace --stats
  pass other/trace_c4jrk "python $ROOT/bench.py --lattice jrk --steps 2 --warmup 1 --cpu-seconds 0 --stream 0" --kernel-trace --stats
  pass other/trace_c3 "python $ROOT/bench.py --single --lattice jrk --max-expand 2000000 --steps 1 --warmup 0 --cpu-seconds 0" --kernel-trace --stats
  pass other/trace_c5 "python $ROOT/bench.py --config c5 --steps 3 --warmup 1 --cpu-seconds 0" --kernel-trace --stats
  cd $ROOT
  python profiles/summarize_rocprof.py "$OUT/block" > "$OUT/summary_block.txt" 2>&1
  python profiles/summarize_rocprof.py "$OUT/bulk" > "$OUT/summary_bulk.txt" 2>&1
  python profiles/summarize_rocprof.py "$OUT/other" > "$OUT/summary_other.txt" 2>&1
  find "$OUT" -name "*.db" -delete
  python tools/make_traffic_json.py "$OUT/summary_block.txt" profiles/${TAG}_c4acc_blocking.txt "$OUT/summary_bulk.txt" > $OUT/traffic.json 2> $OUT/traffic.err; head -c 700 $OUT/traffic.json; echo ;;
other)
  timeout 100 python bench.py --config c5 --steps 3 --warmup 1 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; head -c 300 $OUT/bench_c5.json; echo
  timeout 60 python bench.py --config lpa --steps 2 --warmup 1 > $OUT/bench_lpa.json 2> $OUT/bench_lpa.err; head -c 300 $OUT/bench_lpa.json; echo
  timeout 60 python bench.py --config plpa --steps 3 --warmup 1 > $OUT/bench_plpa.json 2> $OUT/bench_plpa.err; head -c 300 $OUT/bench_plpa.json; echo
  timeout 120 python bench.py --single --lattice jrk --steps 1 --warmup 1 --cpu-seconds 5 > $OUT/bench_c3.json 2> $OUT/bench_c3.err; head -c 300 $OUT/bench_c3.json; echo
  timeout 120 python bench.py --lattice jrk --steps 5 --warmup 2 --cpu-seconds 8 --stream 0 > $OUT/bench_c4jrk.json 2> $OUT/bench_c4jrk.err; head -c 300 $OUT/bench_c4jrk.json; echo
  # the C4-JRK batch at the survey's cap (2 000 000 per query): recycled pools of 48 leading workgroups, launches of 96 queries
  # (the epoch-tagged table holds what a launch creates), a CPU replay of every 64-th query at the full cap
  MPLX_DEADLINE_S=600 timeout 1200 python tools/c4jrk_full_cap.py $OUT/bench_c4jrk_2m.json > $OUT/bench_c4jrk_2m.out 2> $OUT/bench_c4jrk_2m.err; head -c 600 $OUT/bench_c4jrk_2m.out; echo; tail -3 $OUT/bench_c4jrk_2m.err ;;
phase)
  for m in tail bulk; do
    MPLX_LIB=$ROOT/build_tmp/libmplx_timers.so timeout 200 python tools/phase_table.py run $m > $OUT/phase_$m.raw 2> $OUT/phase_$m.err
    python tools/phase_table.py parse $OUT/phase_$m.raw > $OUT/phase_$m.txt 2>&1; tail -3 $OUT/phase_$m.txt
    gzip -f $OUT/phase_$m.raw
  done ;;
replay)
  timeout 200 python tools/full_parity_c4.py $OUT/full_parity_c4jrk.json jrk > $OUT/full_parity_jrk.log 2>&1; tail -n 1 $OUT/full_parity_jrk.log | cut -c1-500
  timeout 420 python tools/full_parity_c4.py $OUT/full_parity_c4.json > $OUT/full_parity.log 2>&1; tail -n 1 $OUT/full_parity.log | cut -c1-500 ;;
esac
done
