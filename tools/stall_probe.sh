# Fresh bench processes back to back with the launch trace on: looks for launches whose kernel time is far off.
# usage (GPU box): bash tools/stall_probe.sh [n_processes] [out_dir]
N=${1:-8}
OUT=${2:-gpurun_out/stall_probe}
mkdir -p $OUT
for i in $(seq 1 $N); do
  MPLX_BENCH_TRACE=1 timeout 150 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/p$i.out 2> $OUT/p$i.err
done
grep -h "kernel\|Gcycles" $OUT/p*.err | cut -c1-250
