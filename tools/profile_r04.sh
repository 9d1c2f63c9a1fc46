#!/bin/bash
# Round-4 evidence on a GPU box (repo root): rocprofv3 passes of (a) the default blocking C4-ACC step and (b) the streamed leg
# (two lanes in flight).  Counters in their own passes (no tracing with --pmc).  Summaries land in gpurun_out/<tag>/ -- copy
# into profiles/.  usage: tools/profile_r04.sh <tag> [stream flags...]
set -u
TAG=${1:-r04p}; shift || true
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT/block" "$OUT/stream"
export TMPDIR=/tmp
ROOT=$PWD
BLOCK="python $ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --stream 0"
STREAM="python $ROOT/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --stream 6 $*"
cd /tmp
pass() { d=$1; shift; what=$1; shift; timeout 150 rocprofv3 "$@" -d "$OUT/$d" -o bench -- $what > "$OUT/$d.log" 2>&1; tail -n 1 "$OUT/$d.log" | cut -c1-200; }
pass block/trace "$BLOCK" --kernel-trace --stats
pass block/pmc_fetch "$BLOCK" --pmc FETCH_SIZE
pass block/pmc_write "$BLOCK" --pmc WRITE_SIZE
pass block/pmc_sq1 "$BLOCK" --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES
pass block/pmc_sq2 "$BLOCK" --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU
pass stream/trace "$STREAM" --kernel-trace --stats
pass stream/pmc_fetch "$STREAM" --pmc FETCH_SIZE
pass stream/pmc_write "$STREAM" --pmc WRITE_SIZE
pass stream/pmc_sq1 "$STREAM" --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES
# (c) the bulk phase alone: every query capped at 20 000 expansions, no helper workgroups -- 256 compute units busy, no tail
BULK="python $ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --stream 0 --max-expand 20000 --helpers 0"
mkdir -p "$OUT/bulk"
pass bulk/pmc_sq1 "$BULK" --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES
pass bulk/pmc_sq2 "$BULK" --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU
cd - > /dev/null
python profiles/summarize_rocprof.py "$OUT/bulk" > "$OUT/summary_bulk.txt" 2>&1
python profiles/summarize_rocprof.py "$OUT/block" > "$OUT/summary_block.txt" 2>&1
python profiles/summarize_rocprof.py "$OUT/stream" > "$OUT/summary_stream.txt" 2>&1
find "$OUT" -name "*.db" -delete
grep -c astar_spec "$OUT/summary_block.txt" "$OUT/summary_stream.txt"
