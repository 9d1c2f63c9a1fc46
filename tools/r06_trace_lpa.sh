OUT=$PWD/gpurun_out/r06u2; mkdir -p $OUT; ROOT=$PWD; export TMPDIR=/tmp MPLX_DEADLINE_S=120
cd /tmp
timeout 170 rocprofv3 --kernel-trace --stats -d $OUT/trace_plpa -o bench -- python $ROOT/bench.py --config plpa --steps 3 --warmup 1 --cpu-seconds 0 > $OUT/trace_plpa.log 2>&1; tail -1 $OUT/trace_plpa.log | cut -c1-200
timeout 170 rocprofv3 --kernel-trace --stats -d $OUT/trace_lpa -o bench -- python $ROOT/bench.py --config lpa --map 256 --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/trace_lpa.log 2>&1; tail -1 $OUT/trace_lpa.log | cut -c1-200
cd $ROOT
python profiles/summarize_rocprof.py $OUT > $OUT/summary.txt 2>&1; find $OUT -name "*.db" -delete; head -60 $OUT/summary.txt | cut -c1-200
