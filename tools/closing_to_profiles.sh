#!/bin/bash
# copies what a closing run (tools/r06_final.sh <tag>) left in gpurun_out/<tag> to profiles/<tag>_* under the names DESIGN.md 5 cites
# usage: tools/closing_to_profiles.sh r06ag
set -u
T=$1; G=gpurun_out/$T; P=profiles
cp $G/pytest_gpu.txt $P/${T}_pytest_gpu.txt
for k in default c3 c4jrk c4jrk_2m c5 lpa plpa; do [ -s $G/bench_$k.json ] && cp $G/bench_$k.json $P/${T}_bench_$k.json; done
[ -s $G/summary_block.txt ] && cp $G/summary_block.txt $P/${T}_c4acc_blocking.txt
[ -s $G/summary_bulk.txt ] && cp $G/summary_bulk.txt $P/${T}_c4acc_bulk_capped20k.txt
[ -s $G/summary_other.txt ] && cp $G/summary_other.txt $P/${T}_traces_stream_c4jrk_c3_c5.txt
for m in tail bulk; do [ -s $G/phase_$m.txt ] && cp $G/phase_$m.txt $P/${T}_phase_table_$m.txt; done
for k in c4 c4jrk; do [ -s $G/full_parity_$k.json ] && cp $G/full_parity_$k.json $P/${T}_full_parity_$k.json; done
if [ -s $G/summary_block.txt ]; then python tools/make_traffic_json.py $P/${T}_c4acc_blocking.txt $P/${T}_c4acc_blocking.txt $P/${T}_c4acc_bulk_capped20k.txt > $P/traffic.json; head -c 400 $P/traffic.json; echo; fi
ls -la $P/${T}_*
