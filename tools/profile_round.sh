#!/bin/bash
# One round's evidence on a GPU box: the C4 passes of tools/profile_c4.sh plus kernel traces of the C5 tick
# (astar_poly_kernel) and of the C3 single query (astar_spec_kernel<128,4,JRK,...,help>).  Summaries land in
# gpurun_out/<tag>/ (copy into profiles/).   usage (repo root): tools/profile_round.sh <tag>
set -u
TAG=${1:-r03}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
tools/profile_c4.sh "$TAG/c4"
ROOT=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/c5/trace" -o bench -- python $ROOT/bench.py --config c5 --steps 3 --warmup 1 --cpu-seconds 0 > "$OUT/bench_c5_trace.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/c3/trace" -o bench -- python $ROOT/bench.py --single --lattice jrk --max-expand 2000000 --steps 1 --warmup 0 --cpu-seconds 0 > "$OUT/bench_c3_trace.log" 2>&1
cd - > /dev/null
python profiles/summarize_rocprof.py "$OUT/c5" > "$OUT/summary_c5.txt" 2>&1
python profiles/summarize_rocprof.py "$OUT/c3" > "$OUT/summary_c3.txt" 2>&1
find "$OUT" -name "*.db" -delete
tail -n 1 "$OUT/bench_c5_trace.log" | cut -c1-400
tail -n 1 "$OUT/bench_c3_trace.log" | cut -c1-400
