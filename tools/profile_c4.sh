#!/bin/bash
# rocprofv3 passes of the default bench workload (C4-ACC, 1024 queries, helpers on), one launch each:
#   kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in separate --pmc passes (no tracing with counters),
#   then SQ wave counters.  Summaries land in gpurun_out/<tag>/summary.txt (copy into profiles/).
# usage (GPU box, repo root): tools/profile_c4.sh <tag> [extra bench.py flags]
set -u
TAG=${1:-r02p}; shift || true
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 1 --warmup 0 --cpu-seconds 0 $*"
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- $BENCH > "$OUT/bench_trace.log" 2>&1
timeout 240 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o bench -- $BENCH > "$OUT/bench_fetch.log" 2>&1
timeout 240 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o bench -- $BENCH > "$OUT/bench_write.log" 2>&1
timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES -d "$OUT/pmc_sq1" -o bench -- $BENCH > "$OUT/bench_sq1.log" 2>&1
timeout 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU -d "$OUT/pmc_sq2" -o bench -- $BENCH > "$OUT/bench_sq2.log" 2>&1
cd - > /dev/null
python profiles/summarize_rocprof.py "$OUT" > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*.db" -delete   # the databases are large; the summary is what is kept
for f in "$OUT"/bench_*.log; do tail -n 2 "$f" | cut -c1-300; done
