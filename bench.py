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

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

_T0 = time.perf_counter()


def _log(msg):
    """Progress on stderr (stdout carries the ONE JSON line): where a run that is cut off by an outer timeout had got to."""
    print(f"[bench +{time.perf_counter() - _T0:7.1f} s] {msg}", file=sys.stderr, flush=True)


HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured)


def algorithmic_bytes(control, n_expanded, voxel_reads, n_succ_finite):
    """B_exp = S_in + R_vox + N_succ (S_out + S_probe), summed over the run."""
    ns = {3: 6, 7: 9}[control]
    s_state = 8 * ns
    s_in = s_state + 16
    s_out = s_state + 8 + 4 + 4
    s_probe = 2 * ns * 4 + 8
    return n_expanded * s_in + voxel_reads + n_succ_finite * (s_out + s_probe)


def main():
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
    ap.add_argument("--config", choices=["c4", "c5", "lpa"], default="c4",
                    help="c4 (default): the query batch on the voxel map; c5: BASELINE config 5 -- one decentralised replanning tick of 16 robots "
                         "(Team2) through the moving-obstacle planner, batched in one launch; lpa: the replanning cycle of map_replanner_node.cpp "
                         "(plan, obstacle on the path -> updateBlockedNodes, removed -> updateClearedNodes, getSubStateSpace(1)) on the --map^3 map, "
                         "LPA* repair time next to a fresh device A*")
    ap.add_argument("--c5-capped", action="store_true",
                    help="--config c5: round 3's variant (distance heuristic, max_expand 20000) instead of the reference's planner parameters")
    ap.add_argument("--no-throughput", action="store_true",
                    help="N > 1, strong scaling: skip the additional throughput phase (a 1024 x N query stream through the same sharded path)")
    ap.add_argument("--two-per-cu", action="store_true",
                    help="measurement only (DESIGN.md, round 4): two 256-lane workgroups of eight expansion units per compute unit instead of one 512-lane workgroup "
                         "of sixteen, without helper workgroups (use with --helpers 0 --max-expand 20000 --stream 0): no gain, not a product configuration")
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
    args = ap.parse_args()
    if args.single:
        args.queries = 1
    if args.config == "c5":
        return bench_c5(args)
    if args.config == "lpa":
        return bench_lpa(args)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # MPLX_BENCH_FORCE_DIST=1: take the multi-rank code path (process group, RCCL broadcast of the map, run_sharded, gather)
    # even with ONE rank -- the only way to run that path over the real "nccl" back-end on a one-GPU box (RCCL refuses two
    # ranks on one device; tests/test_bench_multirank.py uses it next to the 2-rank gloo dry run)
    multi = world > 1 or os.environ.get("MPLX_BENCH_FORCE_DIST") == "1"
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # (dry runs of the multi-rank path on a one-GPU box: MPLX_BENCH_SHARE_GPU=1 puts every rank on device 0 and
    #  MPLX_BENCH_BACKEND=gloo replaces RCCL, which refuses two ranks on one device; small collectives then run on the host)
    backend = os.environ.get("MPLX_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("MPLX_BENCH_SHARE_GPU") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from mpl_ros_amd import dist as mdist
    from mpl_ros_amd import mapgen
    from mpl_ros_amd.planner import ACC, JRK, VoxelMapPlanner, VoxelMapUtil, Waypoint3D

    control = ACC if args.lattice == "acc" else JRK
    n = args.map
    res = 0.1
    origin = (0.0, 0.0, 0.0)

    # ---- map: rank 0 generates, RCCL broadcast puts one replica in every GPU's HBM
    t0 = time.time()
    meta = torch.zeros(7, dtype=torch.float64, device=dev)
    if rank == 0:
        grid, _, _, _, _, _ = mapgen.benchmark_map(n)
        map_t = torch.from_numpy(grid.reshape(-1)).to(dev)
        meta[:] = torch.tensor([n, n, n, *origin, res], dtype=torch.float64)
    else:
        grid = None
        map_t = torch.empty(n * n * n, dtype=torch.int8, device=dev)
    t_gen = time.time() - t0
    torch.cuda.synchronize()
    t0 = time.time()
    if multi:
        mdist.broadcast_map(dist, map_t, meta, src=0)
        torch.cuda.synchronize()
    t_bcast = time.time() - t0
    if grid is None:
        grid = map_t.cpu().numpy().reshape(n, n, n)  # host copy only to draw free query cells

    mu = VoxelMapUtil(dev_index)
    mu.setMapDevice(map_t.data_ptr(), origin, (n, n, n), res)

    # ---- the query stream and this rank's share of it
    if args.single:
        g = {256: 23.55, 512: 49.15}.get(n, round((n - 20) * res, 2) + 0.05)
        queries = [((2.05, 2.05, 2.05), (g, g, g))]  # the bubbles carved by benchmark_map()
        parts = [[0]] + [[] for _ in range(world - 1)]
    elif args.scaling in ("strong", "weak"):  # phase A: ONE stream of `queries` dealt over the ranks (phase B, N > 1: queries x N)
        queries = mapgen.c4_queries(grid, origin, res, args.queries, rank=0)
        parts = mdist.partition(queries, world, args.shard)
    else:
        queries = mapgen.c4_queries(grid, origin, res, args.queries, rank=rank)
        parts = None
    mine = list(range(len(queries))) if parts is None else parts[rank]
    n_local = max(len(mine), 1)

    # ---- planner: C4 parameters
    jrk = control == JRK
    U = mapgen.control_lattice(1.0, 2 if jrk else 1, True)
    if not jrk:
        # BASELINE.md bounds wall time with max_num = 2 000 000 expansions (C3); the same cap is applied to
        # the C4 queries: one of the 1024 random pairs has a goal that is not reachable within it
        max_expand = args.max_expand if args.max_expand > 0 else 2_000_000
        slots = args.slots or 1024
    else:
        max_expand = args.max_expand if args.max_expand > 0 else (2_000_000 if args.single else 20000)
        slots = args.slots or 768
    caps = mapgen.c4_pools(jrk, n_local, max_expand, per_q=args.max_nodes)
    if not jrk and n_local < 1024:  # a small share of a heavy-tailed stream: leave room for its longest queries
        caps = mapgen.c4_pools(jrk, max(n_local, 256), max_expand, per_q=args.max_nodes)
    pl = VoxelMapPlanner(False)
    pl.setMapUtil(mu)
    pl.setVmax(2.0)
    pl.setAmax(1.0)
    if jrk:
        pl.setJmax(1.0)
    pl.setDt(1.0)
    pl.setU(U)
    pl.setTol(0.5)
    pl.setMaxNum(max_expand)
    pl.setCapacity(min(slots, n_local), caps["nodes"], caps["edges"], caps["log"])
    pl.setHelpers(args.helpers, args.help_reserved)
    if args.two_per_cu:
        pl.setSpeculation(82)

    def wp(p):
        w = Waypoint3D(control)
        w.pos = np.array(p, dtype=np.float64)
        return w

    starts = [wp(queries[i][0]) for i in mine]
    goals = [wp(queries[i][1]) for i in mine]

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    state = {"results": [], "kernel_ms": 0.0, "merged": None, "per_rank": None}

    def plan_fn(indices):
        """This rank's share of one step (mdist.run_sharded hands it parts[rank]); returns the result rows."""
        assert list(indices) == mine
        res = pl.planBatch(starts, goals) if mine else []
        state["results"] = res
        if mine:
            state["kernel_ms"] += pl.lastKernelMs()
        return [mdist.result_row(qi, r.status, r.n_expanded, r.n_nodes, r.cost, r.expand_hash, r.traj_len) for qi, r in zip(mine, res)]

    sharded = multi and parts is not None

    def step():
        if sharded:  # the function tests/test_multiproc_gloo.py drives with gloo: partition -> plan -> gather -> merge
            state["merged"], _, state["per_rank"] = mdist.run_sharded(dist, torch, rank, world, queries, plan_fn, mode=args.shard,
                                                                      device=coll_dev, sync=torch.cuda.synchronize)
        else:
            plan_fn(mine)

    if os.environ.get("MPLX_BENCH_TRACE"):
        import faulthandler
        import signal
        faulthandler.register(signal.SIGUSR1, all_threads=True)
    if rank == 0:
        _log(f"map and planner ready; {args.warmup} warm-up + {args.steps} timed steps of {len(mine)} queries")
    for _ in range(args.warmup):
        step()
        if os.environ.get("MPLX_BENCH_TRACE"):
            Tq = np.array([pl.queryTiming(k) for k in range(len(mine))])
            late = np.argsort(-Tq[:, 1])[:3]
            print(f"[trace] warmup step done, kernel {pl.lastKernelMs():.0f} ms {pl.helperStats()} last start {Tq[:, 0].max():.2f} s; running at 2 s / 4 s / 6 s: "
                  f"{int(((Tq[:, 0] <= 2) & (Tq[:, 1] > 2)).sum())} / {int(((Tq[:, 0] <= 4) & (Tq[:, 1] > 4)).sum())} / {int(((Tq[:, 0] <= 6) & (Tq[:, 1] > 6)).sum())}; latest "
                  f"{[(int(k), round(float(Tq[k, 0]), 2), round(float(Tq[k, 1]), 2), int(state['results'][k].n_expanded), int(Tq[k, 2])) for k in late]}", file=sys.stderr, flush=True)
            if pl.lastKernelMs() > 5000:  # a stalled launch: where did the latest queries spend their cycles (Gcycles per phase)
                for k in late:
                    print(f"[trace]   q {int(k)} Gcycles {({n: round(v / 1e9, 2) for n, v in pl.queryCycles(int(k)).items()})}", file=sys.stderr, flush=True)
    barrier()
    state["kernel_ms"] = 0.0
    t0 = time.perf_counter()
    for i_step in range(args.steps):
        step()
        if rank == 0 and (i_step + 1) % 5 == 0:
            _log(f"step {i_step + 1} of {args.steps}")
        if os.environ.get("MPLX_BENCH_TRACE"):
            Tq = np.array([pl.queryTiming(k) for k in range(len(mine))])
            late = np.argsort(-Tq[:, 1])[:4]
            print(f"[trace] step done, kernel {pl.lastKernelMs():.0f} ms {pl.helperStats()} latest (q, begin, end, expansions, slot): "
                  f"{[(int(k), round(Tq[k, 0], 2), round(Tq[k, 1], 2), int(state['results'][k].n_expanded), int(Tq[k, 2])) for k in late]}", file=sys.stderr, flush=True)
    barrier()
    elapsed = time.perf_counter() - t0
    local_s = elapsed
    if rank == 0:
        _log(f"timed steps done: {1e3 * elapsed / args.steps:.1f} ms per step")
    results, kernel_ms = state["results"], state["kernel_ms"]

    n_exp = sum(r.n_expanded for r in results)
    reads = sum(r.voxel_reads for r in results)
    nsf = sum(r.n_succ_finite for r in results)
    status = np.bincount(np.array([r.status for r in results], dtype=np.int64), minlength=7)[:7]
    lat = np.array([pl.queryTiming(k)[1] - pl.queryTiming(k)[0] for k in range(len(mine))]) if mine else np.zeros(0)
    per_rank = [[local_s, float(n_exp), float(len(mine))]]
    longest_ms = float(lat.max()) * 1e3 if len(lat) else 0.0
    if multi:
        lm = torch.tensor([longest_ms], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(lm, op=dist.ReduceOp.MAX)
        longest_ms = float(lm.item())
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([n_exp, reads, nsf] + status.tolist(), dtype=torch.int64, device=coll_dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        tot_exp = int(c[0].item())
        tot_status = c[3:].tolist()
        st = torch.tensor([local_s, float(n_exp), float(len(mine))], dtype=torch.float64, device=coll_dev)
        sts = [torch.empty_like(st) for _ in range(world)]
        dist.all_gather(sts, st)
        per_rank = [s.cpu().tolist() for s in sts]
        if sharded:  # the whole stream's rows, merged in stream order by run_sharded: every query exactly once
            assert sum(int(m[2]) for m in state["merged"]) == tot_exp
    else:
        tot_exp = n_exp
        tot_status = status.tolist()

    if args.dump_queries and rank == 0 and mine:
        T = [pl.queryTiming(k) for k in range(len(mine))]
        json.dump({"query": mine, "n_expanded": [int(r.n_expanded) for r in results], "status": [int(r.status) for r in results],
                   "t_begin": [t[0] for t in T], "t_end": [t[1] for t in T], "slot": [t[2] for t in T],
                   "n_nodes": [int(r.n_nodes) for r in results], "kernel_ms": pl.lastKernelMs(),
                   "cycles": {int(k): pl.queryCycles(int(k)) for k in np.argsort([-r.n_expanded for r in results])[:16]}}, open(args.dump_queries, "w"))

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = tot_exp * args.steps / elapsed
        k_ms = kernel_ms / args.steps  # rank 0's astar kernel, HIP events on its launch stream
        alg = algorithmic_bytes(control, n_exp, reads, nsf)
        achieved = alg / (k_ms * 1e-3) / 1e9
        lattice = args.lattice.upper()
        if args.single:
            workload = f"C3-{lattice}: single query (2.05,..)->({queries[0][1][0]},..) on a "
        elif args.scaling == "strong" or not multi:
            workload = f"C4-{lattice}: {len(queries)} independent start/goal queries sharded over {world} GPU(s) ({args.shard}) on one shared "
        elif args.scaling == "weak":
            workload = f"C4-{lattice}: a stream of {len(queries)} x {world} independent start/goal queries dealt over {world} GPUs ({args.shard}; {len(queries)} per GPU) on one shared "
        else:
            workload = f"C4-{lattice}: {len(queries)} independent start/goal queries per GPU on one shared "
        out = {
            "metric": "node_expansions_per_s",
            "value": value,
            "unit": "expansions/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong" if (args.scaling == "strong" or args.single) else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": workload + f"{n}^3 random-box voxel map (10% occupied, seed 20250620), {U.shape[0]}-primitive {args.lattice} lattice, "
                            f"dt 1 v_max 2 a_max 1 tol 0.5" + (f", max_expand {max_expand}" if max_expand > 0 else ""),
                "queries_total": len(queries) * (world if args.scaling == "weak-only" and not args.single else 1),
                "queries_rank0": len(mine),
                "map_dim": [n, n, n],
                "n_primitives": int(U.shape[0]),
                "slots_per_gpu": min(slots, n_local),
                "helpers": {"per_leader": args.helpers, "reserved": args.help_reserved, **(pl.helperStats() if mine else {})},
                "parallelism": f"queries sharded, {world} map replica(s), RCCL broadcast",
            },
            "expansions_per_step": tot_exp,
            # what rank 0's searches did in the last step (the HBM traffic accounted by structure)
            "search_counters_rank0": {k: int(sum(getattr(r, k) for r in results)) for k in
                                      ("n_expanded", "n_nodes", "n_edges", "n_succ", "n_succ_finite", "voxel_reads", "n_push", "n_reopen", "n_refill", "n_evict")},
            "plan_status_counts": {"ok": tot_status[0], "no_path": tot_status[1], "start_occupied": tot_status[2],
                                   "max_expand": tot_status[3], "pool_full": tot_status[4], "internal": tot_status[5],
                                   "traj_too_long": tot_status[6]},
            # real per-query plan() latency on the device clock (query picked up by a workgroup -> result written),
            # rank 0's share of the last step; the batch itself takes ms_per_step
            "plan_latency_ms": {"p50": float(np.percentile(lat, 50)) * 1e3, "p90": float(np.percentile(lat, 90)) * 1e3,
                                "p99": float(np.percentile(lat, 99)) * 1e3, "max": float(lat.max()) * 1e3, "mean": float(lat.mean()) * 1e3},
            "map_setup_s": {"generate": round(t_gen, 3), "rccl_broadcast": round(t_bcast, 4)},
            "per_rank": [{"rank": r, "seconds_per_step": round(p[0] / args.steps, 4), "expansions_per_step": int(p[1]), "queries": int(p[2])} for r, p in enumerate(per_rank)],
            # a query is a serial pop chain and never spans GPUs: however the stream is dealt, a step cannot end before its
            # longest query does (device clock, last step, max over the ranks) -- the floor of the strong-scaling line
            "tail_bound": {"longest_query_ms": longest_ms, "note": "strong scaling of ONE 1024-query stream is bounded below by the longest query alone; "
                                                                      "query throughput over N GPUs is in `throughput` (N > 1)"},
            "roofline": {"bound": "hbm", "limiter": "latency (serial pop -> look-up -> commit chain of the longest query; see roofline.valu)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": pl.kernelName(), "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg,
                         "bytes_per_expansion": alg / max(n_exp, 1), "launch": "rank 0's share of the stream"},
        }
        # HBM traffic of the same launch from the committed rocprofv3 PMC passes (tools/profile_c4.sh; counters
        # cannot be collected inside this process); only attached when the profile is of this workload
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            # ... and of this BINARY: the committed passes' kernel time (rocprofv3 --kernel-trace) must agree with the kernel time
            # measured here within 3 %, or the counters are not attached (round 4 shipped counters of the kernel before its last fix)
            agree = abs(tr["kernel_ms_trace"] - k_ms) <= 0.03 * k_ms
            if args.lattice == "acc" and len(mine) == 1024 and n == 512 and not args.single and not agree:
                out["roofline"]["traffic_refused"] = (f"profiles/traffic.json is of a kernel that takes {tr['kernel_ms_trace']:.0f} ms per launch, this run measured "
                                                      f"{k_ms:.0f} ms: not the same binary / machine state, counters not attached")
            if args.lattice == "acc" and len(mine) == 1024 and n == 512 and not args.single and agree:
                out["roofline"]["traffic"] = (tr["FETCH_SIZE_KB"] + tr["WRITE_SIZE_KB"]) * 1024.0
                out["roofline"]["traffic_source"] = tr["profile"]
                out["roofline"]["traffic_kernel_ms_trace"] = tr["kernel_ms_trace"]
                if tr.get("SQ_INSTS_VALU"):
                    # the NEARER ceiling of this kernel is VALU issue, not HBM: a wave64 VALU instruction occupies its SIMD for
                    # 4 cycles, the machine has 1024 SIMDs at 2.4 GHz; insts from the same committed counter passes
                    insts = float(tr["SQ_INSTS_VALU"])
                    floor_s = insts * 4.0 / (1024 * 2.4e9)
                    out["roofline"]["valu"] = {"insts": insts, "insts_per_expansion": insts / max(tr.get("expansions", n_exp), 1), "floor_s": floor_s,
                                               "frac": floor_s / (k_ms * 1e-3), "wait_frac": tr.get("SQ_WAIT_ANY_over_WAVE_CYCLES"),
                                               "note": "fraction of the launch the VALU instruction stream alone would take at full issue on every SIMD; "
                                                       "both this and the HBM fraction are low: the launch is latency-bound (serial pop chain per query)"}
        except Exception:
            pass
        if args.cpu_seconds > 0 and mine and world == 1:  # (the CPU baseline is a rank-0, N = 1 leg)
            # a single capped query is sampled on the CPU with a smaller cap; the GPU then repeats the query with
            # that cap (untimed) so that the parity check compares equal searches
            cpu_cap = min(max_expand, 250_000) if (args.single and max_expand > 0) else max_expand
            par_results, par_traj = results, lambda k: pl.getTraj(k)
            if cpu_cap != max_expand:
                pl.setMaxNum(cpu_cap)
                par_results = pl.planBatch(starts, goals)
            gpu_exp = [r.n_expanded for r in par_results]
            _log(f"CPU baseline leg (budget {args.cpu_seconds:.0f} s)")
            nthr = 1 if args.single else (args.cpu_threads if args.cpu_threads > 0 else min(os.cpu_count() or 1, 64))
            try:
                out["cpu_baseline"] = cpu_baseline(grid, origin, res, control, U, cpu_cap, [queries[i] for i in mine], gpu_exp, args.cpu_seconds, nthr)
            except Exception as e:  # (a worker process that died, no gcc ...: the GPU line is still printed, the failure is named)
                out["cpu_baseline"] = {"value": None, "unit": "expansions/s", "cores": nthr, "kind": "port", "sample": "", "error": f"{type(e).__name__}: {e}", "_per_query": {}}
            # the CPU sample doubles as a full-size parity check of the GPU results (checker only): expansion
            # order hash, states created, path cost and the path's actions of every sampled query must be identical
            pq = out["cpu_baseline"].pop("_per_query")
            bad = []
            for k, (ne, nn, cost, h, actions) in pq.items():
                r = par_results[k]
                ok = ne == r.n_expanded and nn == r.n_nodes and h == r.expand_hash
                ok = ok and (cost == r.cost or (np.isinf(r.cost) and not np.isfinite(cost)))
                if ok and actions is not None:
                    ok = np.array_equal(par_traj(k).actions, actions)
                if not ok:
                    bad.append(k)
            out["parity_sample"] = {"queries": len(pq), "mismatches": len(bad), "checked": "expand_hash, n_expanded, n_nodes, cost (bit-exact f64), actions"}
            if bad:
                out["parity_sample"]["first_bad_query"] = int(bad[0])
            _log(f"CPU baseline done: {(out['cpu_baseline']['value'] or 0.0) / 1e6:.2f} M expansions/s on {nthr} cores; parity {len(pq)} queries, {len(bad)} mismatches")
            if cpu_cap != max_expand:
                out["cpu_baseline"]["sample"] += f"; CPU run and the GPU parity run capped at {cpu_cap} expansions"
        # the streamed leg comes last: it frees the blocking leg's pools, and if a lane stops answering (deadline in stream_leg)
        # the line is printed with what the blocking and CPU legs measured and the process leaves without waiting for the device
        n_stream = args.stream if args.stream >= 0 else max(args.steps, 6)
        if n_stream > 0 and world == 1 and not multi and not args.single and mine:
            _log(f"streamed leg: {n_stream} batches, {args.stream_depth} in flight")
            try:
                out["stream"] = stream_leg(args, pl, starts, goals, results, n_stream, control, jrk, max_expand, alg)
                _log(f"streamed leg done: {out['stream']['value'] / 1e6:.1f} M expansions/s, {out['stream']['parity']['mismatches_vs_blocking_step']} mismatches")
            except StreamStalled as e:
                out["stream"] = {"error": f"{e}", "stalled": True}
                _log(f"streamed leg STALLED: {e}")
                print(json.dumps(out), flush=True)
                sys.stderr.flush()
                os._exit(0)  # (a launch that never ends cannot be cancelled; do not let interpreter teardown wait on it)
            except Exception as e:  # (e.g. the lanes' pools do not fit next to something else on the device: the blocking line stands on its own)
                out["stream"] = {"error": f"{type(e).__name__}: {e}"}
                _log(f"streamed leg failed: {out['stream']['error']}")
            try:  # HBM traffic per streamed launch from the committed counter passes of the same leg (tools/profile_r04.sh)
                trs = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get("stream")
                if trs and args.lattice == "acc" and len(mine) == 1024 and n == 512 and args.stream_split == 1:
                    out["stream"]["roofline"]["traffic_per_batch"] = (trs["FETCH_SIZE_KB"] + trs["WRITE_SIZE_KB"]) * 1024.0
                    out["stream"]["roofline"]["traffic_source"] = trs["profile"]
            except Exception:
                pass
    # ---- N > 1, strong scaling: the SAME command also measures query throughput -- a stream of 1024 x N queries dealt by the
    # same run_sharded (every rank then holds what one GPU holds at N = 1).  The strong line above is tail-bound by
    # construction (a query never spans GPUs: its floor is the longest query alone); this one is what "near-linear
    # query-throughput scaling" can be read from.  One JSON line: the throughput figures ride in out["throughput"].
    thr = None
    if multi and sharded and not args.single and not args.no_throughput:
        tq = mapgen.c4_queries(grid, origin, res, args.queries * world, rank=0)
        tparts = mdist.partition(tq, world, args.shard)
        tmine = tparts[rank]
        tcaps = mapgen.c4_pools(jrk, max(len(tmine), 1), max_expand, per_q=args.max_nodes)
        pl.setCapacity(min(slots, max(len(tmine), 1)), tcaps["nodes"], tcaps["edges"], tcaps["log"])
        tstarts = [wp(tq[i][0]) for i in tmine]
        tgoals = [wp(tq[i][1]) for i in tmine]
        tstate = {"kernel_ms": 0.0}

        def tplan(indices):
            assert list(indices) == tmine
            res_t = pl.planBatch(tstarts, tgoals) if tmine else []
            tstate["kernel_ms"] += pl.lastKernelMs() if tmine else 0.0
            return [mdist.result_row(qi, r.status, r.n_expanded, r.n_nodes, r.cost, r.expand_hash, r.traj_len) for qi, r in zip(tmine, res_t)]

        tsteps = args.steps if args.scaling == "weak" else max(1, args.steps // 4)  # (the line's own phase: exactly K steps after W warm-up steps)
        for _ in range(max(1, args.warmup) if args.scaling == "weak" else 1):
            mdist.run_sharded(dist, torch, rank, world, tq, tplan, mode=args.shard, device=coll_dev, sync=torch.cuda.synchronize)  # warm-up
        barrier()
        t0 = time.perf_counter()
        for _ in range(tsteps):
            tmerged, _, tper = mdist.run_sharded(dist, torch, rank, world, tq, tplan, mode=args.shard, device=coll_dev, sync=torch.cuda.synchronize)
        barrier()
        tel = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tel, op=dist.ReduceOp.MAX)
        t_exp = sum(int(m[2]) for m in tmerged)
        thr = {"metric": "node_expansions_per_s", "value": t_exp * tsteps / float(tel.item()), "unit": "expansions/s", "scaling": "weak",
               "queries_total": len(tq), "queries_per_gpu": len(tq) // world, "steps": tsteps, "ms_per_step": 1e3 * float(tel.item()) / tsteps,
               "expansions_per_step": t_exp,
               "per_rank": [{"rank": r, "plan_seconds_last_step": round(p[0], 4), "expansions_per_step": int(p[1])} for r, p in enumerate(tper)]}
    if rank == 0:
        if thr is not None and args.scaling == "weak":
            # the line = the throughput phase (per-GPU work fixed as N grows); the one-stream measurement rides along
            strong = {k: out[k] for k in ("value", "ms_per_step", "expansions_per_step", "per_rank", "tail_bound", "plan_latency_ms", "plan_status_counts") if k in out}
            strong.update({"scaling": "strong", "queries_total": out["config"]["queries_total"], "steps": args.steps,
                           "note": "ONE stream of `queries` dealt over the ranks: bounded below by its longest query (a query never spans GPUs)"})
            out["strong"] = strong
            out.update({"value": thr["value"], "ms_per_step": thr["ms_per_step"], "expansions_per_step": thr["expansions_per_step"], "scaling": "weak",
                        "per_rank": [{"rank": p["rank"], "seconds_per_step": p["plan_seconds_last_step"], "expansions_per_step": p["expansions_per_step"],
                                      "queries": thr["queries_per_gpu"]} for p in thr["per_rank"]]})
            out["config"]["queries_total"] = thr["queries_total"]
            out["config"]["queries_per_gpu"] = thr["queries_per_gpu"]
            for k in ("tail_bound", "plan_latency_ms", "plan_status_counts", "search_counters_rank0", "roofline"):
                if k in out and k != "roofline":
                    out.pop(k)
            if "roofline" in out:
                out["roofline"]["launch"] = "rank 0's share of the ONE-stream phase (out['strong']); the throughput phase launches the same kernel on 1024 queries per GPU"
        elif thr is not None:
            out["throughput"] = thr
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def bench_c5(args):
    """BASELINE config 5: 16-robot decentralised replanning at a fixed 4 s horizon on PolyMapPlanner2D-style moving
    obstacles, one tick (all 16 robots replan) per step, batched in one launch on 1 GPU.  The CPU baseline is the search
    through the REFERENCE's own env_poly_map compiled from where it lies (oracle/_ref/libpolymap_ref.so), one robot
    after the other on one core, like the reference's update_decentralized (robot_team.hpp:60-66)."""
    import torch
    from mpl_ros_amd import poly_map as pm
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    worlds, starts, goals = pm.team2_tick(dt=0.5, t_now=1.0, traj_time=4.0)
    # Planner parameters = the reference's (robot.hpp:109-122: setVmax / setAmax / setDt / setTol(0.5) / setU and nothing else,
    # i.e. the dynamics-aware heuristic and no expansion cap).  --c5-capped: round 3's variant (distance heuristic,
    # max_expand 20 000), which the default run reports as the labelled extra `capped_variant`.
    ref_params = not args.c5_capped
    max_expand = args.max_expand if args.max_expand > 0 else (-1 if ref_params else 20000)
    kw = dict(dt=0.5, v_max=2.0, a_max=1.0, w=10.0)
    team = pm.PolyTeam()
    team.configure(pm.ACC, pm.U9, **kw)
    team.set_worlds(worlds)
    team.set_capacity(16, 1 << 21, 1 << 23, 1 << 22)
    team.set_helpers(args.helpers if args.helpers in (-1, 0) else min(args.helpers, 15))
    world_of = np.arange(16)
    pkw = dict(eps=1.0, tol_pos=0.5, max_expand=max_expand, heur_ignore_dynamics=not ref_params)
    for _ in range(args.warmup):
        team.plan_batch(world_of, starts, goals, **pkw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    for _ in range(args.steps):
        team.set_worlds(worlds)  # a tick re-uploads every robot's obstacle set (the trajectories changed)
        R = team.plan_batch(world_of, starts, goals, **pkw)
        kernel_ms += team.last_kernel_ms()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    n_exp = sum(r.n_expanded for r in R)
    nsf = sum(r.n_succ_finite for r in R)
    n_prims = sum(r.n_succ for r in R)
    # algorithmic bytes per expansion: S_in + obstacle data read by the collision tests of the valid primitives
    # (15 trajectories x (104 B record + 4 hyperplanes x 32 B + 8 segments x 104 B) + the box) + N_succ (S_out + S_probe)
    obs_bytes = sum(104 + 32 * len(o.poly) + 104 * len(o.segs) for o in worlds[0].nonlinear) + 104 + 32 * 4
    alg = n_exp * (48 + 16) + n_prims * obs_bytes + nsf * ((48 + 16) + (2 * 7 * 4 + 8))
    k_ms = kernel_ms / args.steps
    longest = int(np.argmax([r.n_expanded for r in R]))
    cyc = team.cycles(longest)
    per_exp = {k: v / max(R[longest].n_expanded, 1) for k, v in cyc.items()}
    out = {"metric": "node_expansions_per_s", "value": n_exp * args.steps / elapsed, "unit": "expansions/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": "C5: one decentralised replanning tick of the 16 robots of Team2 (robot_team.hpp:275-353), each against the 15 others' "
                                  "trajectories (4 s horizon) + the static box, moving-obstacle planner (env_poly_map), 9-primitive acc lattice, dt 0.5 "
                                  "v_max 2 a_max 1 tol 0.5, " + ("the reference's planner parameters (robot.hpp:109-122): dynamics-aware heuristic, no expansion cap"
                                                                 if ref_params and max_expand <= 0 else
                                                                 f"{'dynamics-aware' if ref_params else 'distance'} heuristic, max_expand {max_expand}") + "; all 16 searches in one launch",
                      "robots": 16, "n_primitives": 9},
           "expansions_per_step": n_exp, "plan_status_counts": {str(k): int(v) for k, v in enumerate(np.bincount([r.status for r in R], minlength=7))},
           "tick_ms": 1e3 * elapsed / args.steps,
           "cycles_per_expansion_longest_robot": {k: v for k, v in per_exp.items() if k != "lookahead_hits"},
           "lookahead": {"helpers_per_robot": team.last_helpers(), "hit_rate_longest_robot": per_exp.get("lookahead_hits", 0.0),
                         "note": "workgroups on the idle compute units run the collision tests of the states a search has just created; identical results"},
           "roofline": {"bound": "hbm", "achieved": alg / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "traffic": None, "kernel": "astar_poly_kernel<64,ACC> (leaders) + astar_poly_kernel<256,ACC> (look-ahead helpers, concurrent launch)", "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg,
                        "note": "16 leader workgroups (one per robot) + helper workgroups on otherwise idle compute units; the obstacle data stays in L2 and the expansion is f64 root solving; latency bound"}}
    if args.cpu_seconds > 0:
        from oracle import refpoly
        if refpoly.available():
            t0 = time.perf_counter()
            n_cpu, bad = 0, 0
            for r in range(16):
                ref = refpoly.RefWorld(worlds[r], pm.ACC, pm.U9, **kw).plan(starts[r], goals[r], eps=1.0, tol_pos=0.5, max_expand=max_expand,
                                                                            heur_ignore_dynamics=not ref_params)
                n_cpu += len(ref["expanded"])
                act, ids, _ = team.traj(r)
                ok = ref["status"] == R[r].status and len(ref["expanded"]) == R[r].n_expanded and ref["n_nodes"] == R[r].n_nodes
                ok = ok and (ref["cost"] == R[r].cost or (np.isinf(ref["cost"]) and np.isinf(R[r].cost)))
                ok = ok and (ref["status"] != 0 or (np.array_equal(act, ref["actions"]) and np.array_equal(ids, ref["node_ids"])))
                bad += 0 if ok else 1
            cpu_s = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": n_cpu / cpu_s, "unit": "expansions/s", "cores": 1, "kind": "reference",
                                   "sample": f"the same tick, the 16 robots one after the other ({n_cpu} expansions, {cpu_s:.1f} s): the reference's env_poly_map "
                                             "compiled from its own headers, driven by the restated best-first loop (GraphSearch is not vendored)",
                                   "tick_ms": 1e3 * cpu_s}
            out["parity_sample"] = {"queries": 16, "mismatches": bad, "checked": "status, n_expanded, n_nodes, cost (bit-exact f64), actions, node ids"}
    if ref_params and args.max_expand <= 0:  # labelled extra: round 3's capped variant of the same tick (not the line's value)
        ckw = dict(eps=1.0, tol_pos=0.5, max_expand=20000, heur_ignore_dynamics=True)
        team.plan_batch(world_of, starts, goals, **ckw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            team.set_worlds(worlds)
            Rc = team.plan_batch(world_of, starts, goals, **ckw)
        torch.cuda.synchronize()
        ce = time.perf_counter() - t0
        out["capped_variant"] = {"note": "NOT the reference's parameters: distance heuristic (setHeurIgnoreDynamics(true)) and max_expand 20 000, round 3's C5 line",
                                 "tick_ms": 1e3 * ce / args.steps, "expansions_per_step": int(sum(r.n_expanded for r in Rc)),
                                 "plan_status_counts": {str(k): int(v) for k, v in enumerate(np.bincount([r.status for r in Rc], minlength=7))}}
    print(json.dumps(out), flush=True)


class StreamStalled(RuntimeError):
    """No ticket of the streamed leg completed within the deadline (the launches cannot be cancelled)."""


def stream_leg(args, pl, starts, goals, ref_results, n_batches, control, jrk, max_expand, alg_bytes_per_batch):
    """Steady-state query throughput with several batches in flight (include/mplx.h mplx_stream_*; north_star: "many independent
    start/goal queries ... shard one-query-per-stream").  The blocking step above lasts as long as its longest query -- one
    serial pop chain on one compute unit -- while most of the machine idles; here `depth` lanes (own HIP stream, own pools)
    share the map replica and batch n + 1's workgroups take the compute units batch n's tail no longer needs.  The same
    1024-query batch is submitted n_batches times; EVERY result of EVERY batch is compared with the blocking step's
    (which the CPU leg below parity-samples).  Reported: expansions/s over the wall time of the whole stream, per-batch
    latency (submit -> done), and the algorithmic-bytes rate."""
    import ctypes as C
    import torch
    from mpl_ros_amd import _capi, mapgen
    nq = len(starts)
    key = lambda r: (r.status, r.traj_len, r.cost, r.n_expanded, r.n_nodes, r.n_edges, r.n_succ_finite, r.voxel_reads, r.expand_hash)
    want = [key(r) for r in ref_results]
    exp_per_batch = sum(r.n_expanded for r in ref_results)
    pl.releasePools()  # the blocking leg's pools (~ 130 GB at C4 size) make room for the lanes'
    depth = max(1, args.stream_depth)
    split = max(1, args.stream_split)
    # the parts of a batch: alternate queries of the launch order (longest straight-line distance first, what the planner sorts
    # by), so that every part is the same mix of long and short queries
    order = sorted(range(nq), key=lambda i: -float(np.sum((starts[i].pos - goals[i].pos) ** 2)))
    parts = [order[k::split] for k in range(split)]
    n_part = max(len(p) for p in parts)
    caps = mapgen.c4_pools(jrk, max(n_part, 256), max_expand, per_q=args.max_nodes or ((420_000 if split == 1 else 450_000) if not jrk else 0))
    st = pl.stream(depth)
    # a lane = one workgroup per compute unit, all of them leading (no reserved helper share unless asked); when a batch's queue
    # is empty at most --stream-helper-limit of its workgroups stay on to help its longest queries, the others exit
    # (diagnostic: MPLX_BENCH_LANE_SLOTS = workgroups of a lane's launch; 128 x 2 lanes = all of them resident at once)
    st.configure(min(n_part, int(os.environ.get("MPLX_BENCH_LANE_SLOTS", "256"))), caps["nodes"], caps["edges"], caps["log"], args.helpers, args.stream_reserved, 1 << 24, args.stream_helper_limit)
    SG = [((_capi.Waypoint * len(p))(*[starts[i].to_c() for i in p]), (_capi.Waypoint * len(p))(*[goals[i].to_c() for i in p]), p) for p in parts]
    mism = 0
    mism_detail = []

    # a batch takes 2 - 4 s at C4 size: a minute without a single completion means a lane no longer answers
    stall_s = float(os.environ.get("MPLX_BENCH_STREAM_STALL_S", "60"))

    def wait_done(t, what):
        t_w = time.perf_counter()
        while not st.done(t):
            if time.perf_counter() - t_w > stall_s:
                raise StreamStalled(f"{what}: ticket {int(t)} not done after {stall_s:.0f} s")
            time.sleep(0.0005)

    def collect(t, part):
        nonlocal mism
        R = st.wait(t)
        for k, r in enumerate(R):
            qi = part[k]
            if key(r) != want[qi]:
                mism += 1
                if len(mism_detail) < 8:
                    mism_detail.append({"ticket": int(t), "query": qi, "got": [float(x) if isinstance(x, float) else int(x) for x in key(r)],
                                        "want": [float(x) if isinstance(x, float) else int(x) for x in want[qi]], "timing": list(st.queryTiming(k))})
        return R

    jobs = [(b, k) for b in range(n_batches) for k in range(split)]  # (batch, part) in submission order
    for t, k in [(st.submit_c(SG[k % split][0], SG[k % split][1], len(SG[k % split][2])), k % split) for k in range(depth)]:  # warm-up: allocates the lanes' pools
        wait_done(t, "warm-up")
        collect(t, SG[k][2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t_progress = t0
    inflight, submitted, kernel_ms = [], 0, []
    first_submit, last_done, left = {}, {}, {b: split for b in range(n_batches)}
    while submitted < len(jobs) or inflight:
        while submitted < len(jobs) and len(inflight) < depth:
            b, k = jobs[submitted]
            now = time.perf_counter()
            first_submit.setdefault(b, now)
            inflight.append((st.submit_c(SG[k][0], SG[k][1], len(SG[k][2])), b, k))
            submitted += 1
        progressed = False
        for item in list(inflight):
            t, b, k = item
            if st.done(t):
                now = time.perf_counter()
                collect(t, SG[k][2])
                kernel_ms.append(st.lastKernelMs())
                left[b] -= 1
                if left[b] == 0:
                    last_done[b] = now
                inflight.remove(item)
                progressed = True
                t_progress = now
        if not progressed:
            if time.perf_counter() - t_progress > stall_s:
                raise StreamStalled(f"{len(last_done)} of {n_batches} batches done, tickets {[int(i[0]) for i in inflight]} in flight: none completed in {stall_s:.0f} s "
                                    f"({mism} mismatches so far)")
            time.sleep(0.0005)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    st.close()
    lat = [last_done[b] - first_submit[b] for b in range(n_batches)]
    gaps = np.diff([0.0] + sorted(last_done[b] - t0 for b in range(n_batches)))
    return {"value": exp_per_batch * n_batches / wall, "unit": "expansions/s", "batches": n_batches, "depth": depth, "split": split, "wall_s": wall,
            "ms_per_batch": 1e3 * wall / n_batches, "steady_state_ms_per_batch": 1e3 * float(np.median(gaps[1:])) if len(gaps) > 2 else None,
            "batch_latency_ms": {"mean": 1e3 * float(np.mean(lat)), "min": 1e3 * float(np.min(lat)), "max": 1e3 * float(np.max(lat))},
            "kernel_ms_per_ticket": {"mean": float(np.mean(kernel_ms)), "max": float(np.max(kernel_ms))},
            "helper_limit": args.stream_helper_limit, "reserved": args.stream_reserved, "kernel": pl.kernelName(),
            "parity": {"batches_checked": n_batches, "warmup_tickets_checked": depth, "queries_per_batch": nq, "mismatches_vs_blocking_step": mism, "mismatch_detail": mism_detail,
                       "checked": "status, traj_len, cost (bit-exact f64), n_expanded, n_nodes, n_edges, n_succ_finite, voxel_reads, expand_hash of every query of every batch"},
            "roofline": {"achieved": alg_bytes_per_batch * n_batches / wall / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg_bytes_per_batch * n_batches / wall / 1e9 / HBM_PEAK_GBS,
                         "note": "algorithmic bytes of all batches / wall time of the stream (launches overlap: a per-launch duration would count shared time twice)"},
            "workload": f"the same {nq}-query batch submitted {n_batches} times" + (f", each as {split} tickets of {n_part} queries (alternate queries of the longest-first order)" if split > 1 else "") +
                        f"; {depth} tickets in flight on {depth} lanes of one map replica (mplx_stream); submit -> done latency per batch beside the throughput"}


def bench_lpa(args):
    """Incremental replanning at BASELINE C2 size: the cycle of map_replanner_node.cpp:175-255 on the 256^3
    random-box map (--map), 27-input lattice.  replan_planner_ (setLPAstar(true): the state space stays in HBM between
    plan() calls) next to planner_ (a fresh A* on the same shared MapUtil, the speculative kernel with its helpers) after
    every step; both costs must agree.  One "step" of the line = one whole cycle; value = the LPA* repair after the obstacle
    landed on the path (kernel ms), the number the replanner exists for."""
    import torch
    from mpl_ros_amd import mapgen
    from mpl_ros_amd.planner import ACC, VoxelMapPlanner, VoxelMapUtil, Waypoint3D
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    n = args.map if args.map != 512 else 256
    grid, origin, res, start, goal, _ = mapgen.benchmark_map(n)
    U = mapgen.control_lattice(1.0, 1, True)
    mu = VoxelMapUtil(0)

    def set_map(g):
        dz, dy, dx = g.shape
        mu.setMap(origin, (dx, dy, dz), g.ravel(), res)

    set_map(grid)

    def planner(lpa):
        pl = VoxelMapPlanner(False)
        pl.setMapUtil(mu)
        pl.setVmax(2.0); pl.setAmax(1.0); pl.setDt(1.0); pl.setU(U); pl.setTol(0.5)
        pl.setCapacity(1, 1 << 19, 1 << 21, 1 << 22)
        pl.setLPAstar(lpa)
        return pl

    def wp(p, v=(0, 0, 0)):
        w = Waypoint3D(ACC)
        w.pos, w.vel = np.array(p, dtype=np.float64), np.array(v, dtype=np.float64)
        return w

    def box_on(grid_now, center, half=2):
        c = [int(round((center[i] - origin[i]) / res - 0.5)) for i in range(3)]  # MapUtil::floatToInt
        cells = []
        for dz in range(-half, half + 1):
            for dy in range(-half, half + 1):
                for dx in range(-half, half + 1):
                    x, y, z = c[0] + dx, c[1] + dy, c[2] + dz
                    if 0 <= x < n and 0 <= y < n and 0 <= z < n and grid_now[z, y, x] == 0:
                        cells.append((x, y, z))
        return cells

    rows = []

    def cycle():
        a, l = planner(False), planner(True)
        s, g = wp(start), wp(goal)
        out = []

        def both(label):
            t0 = time.perf_counter()
            ok_l = l.plan(s, g)
            wl = (time.perf_counter() - t0) * 1e3
            rl, kl = l.getResult(), l.lastKernelMs()
            t0 = time.perf_counter()
            ok_a = a.plan(s, g)
            wa = (time.perf_counter() - t0) * 1e3
            ra, ka = a.getResult(), a.lastKernelMs()
            assert ok_l and ok_a and rl.cost == ra.cost, (label, rl.cost, ra.cost)
            out.append({"step": label, "lpa_ms": kl, "fresh_ms": ka, "lpa_wall_ms": wl, "fresh_wall_ms": wa,
                        "lpa_expansions": int(rl.n_expanded), "fresh_expansions": int(ra.n_expanded), "cost": rl.cost})

        both("first plan")
        tr = l.getTraj()
        wps = tr.getWaypoints()
        cells = box_on(grid, tuple(wps[len(wps) // 2].pos))
        g2 = grid.copy()
        for x, y, z in cells:
            g2[z, y, x] = 100
        set_map(g2)
        t0 = time.perf_counter()
        nb = l.updateBlockedNodes(cells)
        upd_b = (time.perf_counter() - t0) * 1e3
        both("obstacle on the path (updateBlockedNodes)")
        out[-1]["update_ms"], out[-1]["entries_changed"] = upd_b, nb
        set_map(grid)
        t0 = time.perf_counter()
        nc = l.updateClearedNodes(cells)
        upd_c = (time.perf_counter() - t0) * 1e3
        both("obstacle removed (updateClearedNodes)")
        out[-1]["update_ms"], out[-1]["entries_changed"] = upd_c, nc
        tr = l.getTraj()
        t0 = time.perf_counter()
        l.getSubStateSpace(1)
        sub = (time.perf_counter() - t0) * 1e3
        w1 = tr.getWaypoints()[1]
        s = wp(tuple(w1.pos), tuple(w1.vel))
        both("one primitive ahead (getSubStateSpace(1))")
        out[-1]["update_ms"] = sub
        return out

    for _ in range(args.warmup):
        cycle()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rows = cycle()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    rep = rows[1]
    out = {"metric": "plan_wall_time_ms", "value": rep["lpa_ms"], "unit": "ms", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"LPA* replanning cycle of map_replanner_node.cpp:175-255 on the {n}^3 random-box voxel map (BASELINE C2 query, 27-primitive acc lattice, dt 1 "
                                  "v_max 2 a_max 1 tol 0.5): plan, a 5^3-voxel obstacle on the middle of the path, removed again, one primitive ahead; value = kernel ms of the "
                                  "LPA* repair after the obstacle landed; `cycle` lists every step next to a fresh device A* (speculative kernel + helpers) on the same map"},
           "cycle": rows, "lpa_vs_fresh_after_obstacle": rep["fresh_ms"] / max(rep["lpa_ms"], 1e-9)}
    print(json.dumps(out), flush=True)


def _cpu_run(cfg, queries, order, budget_s, procs, caps=None):
    """`procs` worker PROCESSES (oracle/cpu_worker.py), one query at a time each, all mapping ONE read-only copy
    of the voxel map.  A dispatcher thread per worker hands out the next query of `order` until the budget is
    spent; queries still running then are given a grace period and dropped afterwards.  caps: per-query expansion
    cap (query index -> cap) for the queries that are timed over a prefix of their search only."""
    import subprocess
    import threading
    workers = []
    ncpu = os.cpu_count() or 1
    for k in range(procs):
        # pin worker k to its own physical core, every other core when there are enough of them (fewer workers per
        # shared L3); logical CPUs [0, ncpu / 2) are taken to be the first hardware thread of each core
        phys = max(ncpu // 2, 1)
        stride = 2 if procs * 2 <= phys else 1
        cfg_k = dict(cfg, cpu=(k * stride) % phys if procs > 1 else None)
        w = subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "cpu_worker.py"), json.dumps(cfg_k)],
                             stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, bufsize=1)
        workers.append(w)
    for w in workers:
        assert json.loads(w.stdout.readline()).get("ready")
    lock = threading.Lock()
    state = {"next": 0, "n_exp": 0, "nq": 0, "busy": 0.0, "per_query": {}, "lat": [], "last_done": 0.0}
    t_start = time.perf_counter()

    def feed(w):
        while True:
            with lock:
                k = state["next"]
                if k >= len(order) or time.perf_counter() - t_start >= budget_s:
                    return
                state["next"] = k + 1
            i = order[k]
            s, g = queries[i]
            try:
                cap = f" {caps[i]}" if caps and i in caps else ""
                w.stdin.write(f"{i} {s[0]!r} {s[1]!r} {s[2]!r} {g[0]!r} {g[1]!r} {g[2]!r}{cap}\n")
                w.stdin.flush()
                line = w.stdout.readline()
            except (BrokenPipeError, ValueError):
                return
            if not line:
                return  # worker was stopped after the grace period
            r = json.loads(line)
            with lock:
                state["n_exp"] += r["n_expanded"]
                state["nq"] += 1
                state["busy"] += r["seconds"]
                state["lat"].append(r["seconds"])
                state["last_done"] = time.perf_counter() - t_start
                state["per_query"][i] = (r["n_expanded"], r["n_nodes"], r["cost"], r["hash"],
                                         None if r["actions"] is None else np.array(r["actions"], dtype=np.int32))

    ths = [threading.Thread(target=feed, args=(w,), daemon=True) for w in workers]
    for t in ths:
        t.start()
    deadline = t_start + budget_s + max(6.0, 0.5 * budget_s)
    for t in ths:
        t.join(timeout=max(0.0, deadline - time.perf_counter()))
    for w in workers:
        w.kill()
    for t in ths:
        t.join(timeout=5.0)
    state["wall"] = max(state["last_done"], 1e-9)
    return state


def cpu_baseline(grid, origin, res, control, U, max_expand, queries, gpu_expansions, budget_s, procs=1):
    """The CPU oracle (oracle/, a restatement -- kind "port") on a bounded sample of the same queries, built
    -march=native on this host when gcc is present.  Two legs, steady clock around plan() only:
      N processes (one query at a time each; independent queries are the only parallelism the reference offers),
      1 process   (the reference planner's actual mode).
    value = expansions of the N-process sample / its wall time.

    The sample is chosen BEFORE the CPU runs, from the expansion counts the GPU reported, so that it has the batch's own
    mix and not "whatever finished in time": the queries sorted by expansion count, every k-th one taken (k sized to the
    budget at an assumed 2.5e4 expansions/s per core); a sampled query longer than one core can finish within the budget
    is timed over its first L expansions only (the cap is passed to the worker; such a query counts L expansions and is
    left out of the parity check).  Dispatch is longest first."""
    import tempfile
    from oracle import orc
    native = orc.use_native()
    kw = dict(dt=1.0, v_max=2.0, a_max=1.0, tol_pos=0.5, max_expand=max_expand)
    if control == orc.JRK:
        kw["j_max"] = 1.0
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    map_path = os.path.join(shm, f"mplx_bench_map_{os.getpid()}.npy")
    np.save(map_path, np.ascontiguousarray(grid, dtype=np.int8))
    cfg = {"map": map_path, "origin": [float(o) for o in origin], "res": float(res), "control": int(control),
           "U": np.asarray(U, dtype=np.float64).tolist(), "kw": kw, "native": bool(native)}
    try:
        procs = max(1, min(procs, len(queries)))
        order = list(range(len(queries)))
        caps = {}
        if procs > 1:
            rate = 2.5e4
            L = int(rate * budget_s * 0.7)
            by_size = sorted(order, key=lambda i: (gpu_expansions[i], i))
            work = [min(gpu_expansions[i], L) for i in by_size]
            target = rate * budget_s * procs * 0.6
            stride = max(1, int(np.ceil(sum(work) / max(target, 1.0))))
            sample = by_size[stride // 2::stride]
            caps = {i: L for i in sample if gpu_expansions[i] > L and (max_expand <= 0 or L < max_expand)}
            order = sorted(sample, key=lambda i: -min(gpu_expansions[i], L))
        multi = _cpu_run(cfg, queries, order, budget_s * (1.0 if not caps else 4.0), procs, caps)
        per_query = {k: v for k, v in multi["per_query"].items() if k not in caps}
        out = {"value": multi["n_exp"] / multi["wall"], "unit": "expansions/s", "cores": procs, "kind": "port",
               "build": "gcc -O3 -march=native -ffp-contract=off" if native else "gcc -O3 -ffp-contract=off (portable)",
               "value_per_core": multi["n_exp"] / max(multi["busy"], 1e-9),
               "sample": (f"{multi['nq']} of the first {multi['next']} of the {len(queries)} queries of rank 0 completed within the budget " if procs == 1 else
                          f"every {stride}-th of the {len(queries)} queries of rank 0 in order of their expansion count (chosen before the CPU ran: the batch's own mix), "
                          f"{multi['nq']} of {len(order)} completed; the {len(caps)} sampled queries above {L} expansions timed over their first {L} only; ") +
                         f"({multi['n_exp']} expansions, {multi['wall']:.1f} s wall, {multi['busy']:.1f} core-s of plan()); {procs} worker "
                         f"processes, one read-only map shared through /dev/shm",
               "plan_latency_ms": {"p50": 1e3 * float(np.percentile(multi["lat"], 50)) if multi["lat"] else None,
                                   "max": 1e3 * float(np.max(multi["lat"])) if multi["lat"] else None,
                                   "mean": 1e3 * multi["busy"] / max(multi["nq"], 1)}}
        if procs > 1:
            # 1-process leg on queries the N-process leg did not reach, skipping the heavy tail so the leg stays bounded
            done = set(multi["per_query"])
            rest = [i for i in range(len(queries)) if i not in done and gpu_expansions[i] <= 600_000]
            single = _cpu_run(cfg, queries, rest, max(4.0, budget_s * 0.6), 1)
            per_query.update(single["per_query"])
            out["single_thread"] = {"value": single["n_exp"] / max(single["busy"], 1e-9), "cores": 1,
                                    "sample": f"{single['nq']} further queries ({single['n_exp']} expansions, {single['busy']:.1f} s of plan())",
                                    "plan_ms_mean_per_query": 1e3 * single["busy"] / max(single["nq"], 1)}
        out["_per_query"] = per_query
        return out
    finally:
        try:
            os.remove(map_path)
        except OSError:
            pass


if __name__ == "__main__":
    main()
