#!/bin/bash
# Round-6 measurement step (one GPU box): baseline lines of the shipped library, the per-phase tables of the timers build (tail and
# bulk regime), L2 hit-rate counters of the bulk launch.   usage: tools/r06_probe.sh <tag> [steps...]   -> gpurun_out/<tag>/
set -u
TAG=${1:-r06a}; shift
STEPS=${*:-"base bulk phase l2"}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export MPLX_DEADLINE_S=100 TMPDIR=/tmp
ROOT=$PWD
BULK="python $ROOT/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --stream 0 --max-expand 20000 --helpers 0"
for s in $STEPS; do
case $s in
base) timeout 200 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --stream 0 > $OUT/bench_block.json 2> $OUT/bench_block.err
      python -c "import json;d=json.load(open('$OUT/bench_block.json'));print('block', round(d['value']/1e6,2),'M/s', round(d['ms_per_step'],1),'ms', d.get('parity_sample'))" ;;
bulk) timeout 120 $BULK > $OUT/bench_bulk.json 2> $OUT/bench_bulk.err
      python -c "import json;d=json.load(open('$OUT/bench_bulk.json'));print('bulk', round(d['value']/1e6,2),'M/s', round(d['ms_per_step'],1),'ms', d.get('parity_sample'))" ;;
stream) timeout 250 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --stream 8 > $OUT/bench_stream.json 2> $OUT/bench_stream.err
      python -c "import json;d=json.load(open('$OUT/bench_stream.json'));s=d['stream'];print('stream', round(s['value']/1e6,2),'M/s steady', s.get('steady_state_ms_per_batch'), s.get('parity'))" ;;
phase) for m in tail bulk; do
        MPLX_LIB=$ROOT/build_tmp/libmplx_timers.so timeout 200 python tools/phase_table.py run $m > $OUT/phase_$m.raw 2> $OUT/phase_$m.err
        python tools/phase_table.py parse $OUT/phase_$m.raw > $OUT/phase_$m.txt 2>&1; cat $OUT/phase_$m.txt
        gzip -f $OUT/phase_$m.raw
      done ;;
l2) cd /tmp
    for c in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
      n=$(echo $c | tr ' ' '_' | cut -c1-40)
      timeout 170 rocprofv3 --pmc $c -d $OUT/l2_$n -o bench -- $BULK > $OUT/l2_$n.log 2>&1; tail -n 2 $OUT/l2_$n.log | cut -c1-200
    done
    cd $ROOT
    python profiles/summarize_rocprof.py $OUT > $OUT/summary_l2.txt 2>&1; tail -30 $OUT/summary_l2.txt
    find $OUT -name "*.db" -delete ;;
xcd1) # one long query: helpers of the leader's own XCD (33 workgroups launched: blocks 8, 16, 24, 32 share XCD 0 with the leader)
    for v in base x; do for g in 5 33; do
      MPLX_HELP_GRID=$g MPLX_HELP_XCD_STATS=1 MPLX_LIB=$ROOT/build_tmp/libmplx_$v.so timeout 200 python tools/ab.py 2 tail > $OUT/xcd1_${v}_$g.json 2> $OUT/xcd1_${v}_$g.err
      python -c "import json;d=json.load(open('$OUT/xcd1_${v}_$g.json'));print('$v grid $g', d['tail']['kernel_ms'], d['tail']['digests'])"; grep "help xcd" $OUT/xcd1_${v}_$g.err | tail -1
    done; done ;;
ab:*) # ab:<variant>[,<variant>...]:<reps>:<modes separated by +>   e.g. ab:base,e1:3:block+bulk+tail
    IFS=: read _ vars reps modes <<< "$s"
    for v in ${vars//,/ }; do
      lib=$ROOT/build_tmp/libmplx_$v.so; [ $v = product ] && lib=$ROOT/mpl_ros_amd/csrc/libmplx.so
      MPLX_HELP_XCD_STATS=1 MPLX_LIB=$lib timeout 400 python tools/ab.py $reps ${modes//+/ } > $OUT/ab_$v.json 2> $OUT/ab_$v.err
      python - <<PY
import json
try:
    d = json.load(open("$OUT/ab_$v.json"))
    print("$v", {m: (d[m]["min_ms"], d[m]["mean_ms"], d[m]["digests"]) for m in d if m != "lib"})
except Exception as e:
    print("$v failed", e)
PY
      grep "help xcd" $OUT/ab_$v.err | tail -2
    done ;;
esac
done
