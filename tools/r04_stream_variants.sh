#!/bin/bash
# Which ingredient of the streamed leg produces results that differ from the blocking step?  (round 4, build r04z: 6-8 queries of
# 20 batches differ with two lanes in flight; none when rocprofv3 serialises the launches)
set -u
OUT=gpurun_out/${1:-r04v}; mkdir -p $OUT
run() { name=$1; shift; MPLX_BENCH_STREAM_STALL_S=30 timeout 150 python bench.py --steps 1 --warmup 0 --cpu-seconds 0 "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.load(open("$OUT/$name.json")); s=d.get("stream") or {}; p=s.get("parity") or {}
    print("$name: stream", round(s.get("value",0)/1e6,1), "M/s, mismatches", p.get("mismatches_vs_blocking_step"), "of", p.get("batches_checked"), "batches", s.get("error"), [(m["ticket"], m["query"], m["got"][0], m["got"][3]-m["want"][3], m["got"][4]-m["want"][4]) for m in p.get("mismatch_detail", [])])
except Exception as e:
    print("$name failed", e)
PY
}
if [ "${2:-}" = "flags" ]; then
  # MPLX_X_FLAGS diagnostics (mplx_device.h xflags): 1 table probes at agent scope, 2 release / acquire fences around a cache record,
  # 4 alternate halves of a doubled state table
  MPLX_X_FLAGS=1 run probe_agent --stream 14
  MPLX_X_FLAGS=4 run table_halves --stream 14
  MPLX_X_FLAGS=2 run fences --stream 14
  run default --stream 14
  exit 0
fi
if [ "${2:-}" = "oldkernel" ]; then
  # the kernel binary that showed the mismatches (source of commit 7b7b1f7, built under build_tmp/r04w) with a HOST-side change only:
  # does a launch that uses the other half of a doubled state table (no line of the lane's previous launch can be consulted) still differ?
  export MPLX_LIB=$PWD/build_tmp/r04w/libmplx.so
  run old_default --stream 18
  MPLX_X_FLAGS=4 run old_table_halves --stream 18
  exit 0
fi
if [ "${2:-}" = "jitter" ]; then
  # old kernel binary again: (1) blocking batches under a background fill load (tools/r04_jitter_probe.py); (2) lanes of 128 workgroups
  # (both lanes' launches fully resident: no workgroup starts on a compute unit another launch has just left)
  export MPLX_LIB=$PWD/build_tmp/r04w/libmplx.so
  timeout 150 python tools/r04_jitter_probe.py 10 fill > $OUT/jitter_fill.json 2> $OUT/jitter_fill.err; cut -c1-900 $OUT/jitter_fill.json; tail -n 2 $OUT/jitter_fill.err | grep -v amdgpu.ids
  MPLX_BENCH_LANE_SLOTS=128 run old_lanes128 --stream 14
  exit 0
fi
if [ "${2:-}" = "jitter2" ]; then
  # which kind of background load makes the blocking batch differ: ALU-only waves (no memory traffic) or read-only traffic?
  export MPLX_LIB=$PWD/build_tmp/r04w/libmplx.so
  for m in alu read; do timeout 120 python tools/r04_jitter_probe.py 8 $m > $OUT/jitter_$m.json 2> $OUT/jitter_$m.err; cut -c1-700 $OUT/jitter_$m.json; tail -n 2 $OUT/jitter_$m.err | grep -v amdgpu.ids; done
  exit 0
fi
run default --stream 16
run nohelp --stream 12 --helpers 0
run nolimit --stream 12 --stream-helper-limit -1
run depth1 --stream 8 --stream-depth 1
