#!/usr/bin/env python3
"""bench.py -- node-expansions/s of the device-resident A* on the BASELINE.json C4 workload.

One "step" = one pass of the hot path over one batch of queries: `mplx_plan_batch` of the rank's
share of the query stream on the shared 512^3 random-box voxel map (map generator of
C3).  Inputs (map replica, control set) are resident in HBM before the timed region; queries are a few
hundred bytes each.

Multi-GPU (one process per GPU, launched by torch.distributed.run): rank 0 generates the map, one RCCL
broadcast over xGMI puts a replica into every GPU's HBM, no collective on the search path.
  --scaling strong (default, BASELINE config 4): ONE stream of --queries queries, sharded over the ranks
                   (mpl_ros_amd/dist.py: longest-expected-first snake, or q mod nGPU with --shard rr);
                   result rows are gathered and merged on rank 0.
  --scaling weak : every rank plans its own stream of --queries queries.

    python bench.py                       # 1 GPU, defaults
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is the astar kernel: algorithmic bytes (SURVEY.md 8d
B_exp, from the kernel's own counters) / its HIP-event duration; `cpu_baseline` is the CPU oracle
(oracle/, "port") timed on a bounded sample of the same queries on the host's cores (one read-only map
shared by all threads; a 1-thread figure -- the reference planner is single-threaded -- and an N-thread
figure), and the sample doubles as a full-size parity check of the timed GPU results.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# the library's launch deadline is opt-in; a bench run gives every search launch two minutes (a hung launch then ends the line with
# MPLX_ERR_TIMEOUT and the workgroups' watch records instead of the driver's kill)
os.environ.setdefault("MPLX_DEADLINE_S", "120")

from benchmarks import c4, c5, extras, lpa, plpa  # noqa: E402
from benchmarks.common import _log  # noqa: E402


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--queries", type=int, default=1024, help="queries of the stream (strong: in total; weak: per GPU)")
    ap.add_argument("--scaling", choices=["weak", "strong", "weak-only"], default="weak",
                    help="what the line's value is at N > 1.  weak (default): query throughput -- a stream of queries x N dealt over the ranks, every GPU holds "
                         "what one GPU holds at N = 1 (the strong-scaling measurement of the ONE stream of `queries` rides along as out['strong']); strong: the "
                         "other way round (the throughput figures ride in out['throughput']); weak-only: round 1's mode, an independent stream per rank")
    ap.add_argument("--shard", choices=["lpt", "rr"], default="lpt", help="strong scaling: how the stream is dealt to the ranks")
    ap.add_argument("--map", type=int, default=512, help="voxel map edge length")
    ap.add_argument("--lattice", choices=["acc", "jrk"], default="acc")
    ap.add_argument("--single", action="store_true",
                    help="BASELINE.md C3: ONE query (2.05,2.05,2.05)->(49.15,49.15,49.15) on the 512^3 map instead of the C4 batch")
    ap.add_argument("--max-expand", type=int, default=0,
                    help="per-query expansion cap setMaxNum (default: 2 000 000 for acc = the BASELINE.md C3 cap, 20000 for jrk)")
    ap.add_argument("--slots", type=int, default=0)
    ap.add_argument("--max-nodes", type=int, default=0, help="mean states per query used to size the shared pools")
    ap.add_argument("--cpu-seconds", type=float, default=14.0, help="CPU-baseline sample budget of the N-thread leg (0 disables both legs)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="host threads of the N-thread CPU leg (0 = all cores, at most 64)")
    ap.add_argument("--helpers", type=int, default=-1, help="helper workgroups per leading workgroup: -1 auto (4; 2 for the 125-input lattice), 0 off, 2..4")
    ap.add_argument("--help-reserved", type=int, default=-1, help="workgroups that only ever help (-1 auto)")
    ap.add_argument("--config", choices=["c4", "c5", "lpa", "plpa"], default="c4",
                    help="c4 (default): the query batch on the voxel map; c5: BASELINE config 5 -- one decentralised replanning tick of 16 robots "
                         "(Team2) through the moving-obstacle planner, batched in one launch; lpa: the replanning cycle of map_replanner_node.cpp "
                         "(plan, obstacle on the path -> updateBlockedNodes, removed -> updateClearedNodes, getSubStateSpace(1)) on the --map^3 map, "
                         "LPA* repair time next to a fresh device A*; plpa: LPA* on the moving-obstacle planner, the flow of poly_map_replanner_node.cpp "
                         "(updateNodes, plan, getSubStateSpace(1) per replan) next to the batched A* planning afresh")
    ap.add_argument("--c5-capped", action="store_true",
                    help="--config c5: round 3's variant (distance heuristic, max_expand 20000) instead of the reference's planner parameters")
    ap.add_argument("--no-throughput", action="store_true",
                    help="N > 1, strong scaling: skip the additional throughput phase (a 1024 x N query stream through the same sharded path)")
    ap.add_argument("--stream", type=int, default=-1,
                    help="N = 1, C4 batch: batches of the additional streamed leg (mplx_stream: two batches in flight on two lanes of the same map replica; "
                         "every result compared with the blocking step's); -1 auto = max(steps, 6), 0 off")
    ap.add_argument("--stream-depth", type=int, default=2, help="lanes of the streamed leg (each holds a batch's pools: two fit 288 GB at C4 size)")
    ap.add_argument("--stream-helper-limit", type=int, default=32,
                    help="streamed leg: workgroups of a batch that stay on as helpers once its queue is empty (the others leave their compute unit to the next batch)")
    ap.add_argument("--stream-split", type=int, default=1,
                    help="streamed leg: every batch is submitted as this many tickets (alternate queries of the longest-first order, so every part holds the same mix); "
                         "the lanes are sized for a part, so --stream-depth can grow with it (2 parts x 4 lanes fit where 1 x 2 do)")
    ap.add_argument("--stream-reserved", type=int, default=0, help="streamed leg: workgroups of every lane's launch that never lead (help from the start)")
    ap.add_argument("--dump-queries", default="", help="write per-query expansions / device timing of the last step to this JSON file")
    ap.add_argument("--warmup-cap", type=int, default=0, help="warm-up steps run with this expansion cap (a long single query is warmed up on a prefix of its search)")
    ap.add_argument("--extras", type=int, default=-1,
                    help="the other BASELINE configurations as extra keys of the C4 line (c2, c3, c5, lpa: each with its own roofline, single-thread "
                         "CPU baseline and parity check); -1 auto = on for the default 1-GPU C4-ACC line, 0 off, 1 on")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.config == "c5":
        print(json.dumps(c5.run(args)), flush=True)
        return
    if args.config == "lpa":
        print(json.dumps(lpa.run(args)), flush=True)
        return
    if args.config == "plpa":
        print(json.dumps(plpa.run(args)), flush=True)
        return
    default_line = (args.gpus == 1 and not args.single and args.lattice == "acc" and args.map == 512 and args.max_expand == 0 and
                    args.queries == 1024 and args.cpu_seconds > 0 and os.environ.get("MPLX_BENCH_FORCE_DIST") != "1")
    want_extras = args.extras == 1 or (args.extras < 0 and default_line)
    if want_extras and args.stream < 0:
        args.stream = min(max(args.steps, 6), 12)  # (room for the extra configurations inside the driver's window)
    out = c4.run(args)
    if out is not None and want_extras:
        t0 = time.perf_counter()
        out.update(extras.run(args))
        _log(f"extra configurations done in {time.perf_counter() - t0:.1f} s")
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
