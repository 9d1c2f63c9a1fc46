"""C4 leg of bench.py (the driver's line): the 1024-query batch on the shared 512^3 map -- blocking step, CPU baseline + parity sample,
streamed leg, and the multi-rank path (one process per GPU, RCCL broadcast of the map, queries sharded, rows gathered)."""
import json
import os
import sys
import time

import numpy as np

from .common import HBM_PEAK_GBS, ROOT, _log, algorithmic_bytes, cpu_baseline


def _speculation(pl, results):
    try:
        sp = [pl.querySpeculation(k) for k in range(len(results))]
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}
    tot = {k: int(sum(s[k] for s in sp)) for k in ("candidates", "stale", "units_expanded", "units_cut")}
    committed = int(sum(r.n_expanded for r in results))
    tot["units_committed"] = committed
    tot["wasted_fraction_of_expanded"] = (tot["units_expanded"] - committed) / max(tot["units_expanded"], 1)
    return tot


def run(args):
    """Returns the line's dict on rank 0 (None on the other ranks)."""
    if args.single:
        args.queries = 1

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
    # pool recycling (round 6): finished queries hand their chunks back, so the pools hold what the CONCURRENTLY running queries need.
    # Measured on the C4-ACC batch (profiles/r06e_*): 358.8 M states created in all, at most 185.5 M held at any one time (final
    # sizes of the queries running together) -- 250 000 per query of the batch instead of 450 000 (256 M states: 1.38 x that peak)
    recycle = not args.single and os.environ.get("MPLX_BENCH_NO_RECYCLE") != "1"
    per_q = args.max_nodes or (250_000 if (recycle and not jrk and n_local >= 1024) else 0)
    caps = mapgen.c4_pools(jrk, n_local, max_expand, per_q=per_q)
    if not jrk and n_local < 1024:  # a small share of a heavy-tailed stream: leave room for its longest queries
        caps = mapgen.c4_pools(jrk, max(n_local, 256), max_expand, per_q=per_q)
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
    if recycle:
        pl.setPoolRecycling(True)

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
    warmup_cap = int(getattr(args, "warmup_cap", 0) or 0)  # (extras: a long single query is warmed up on a prefix of its search)
    if warmup_cap > 0:
        pl.setMaxNum(warmup_cap)
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
    if warmup_cap > 0:
        pl.setMaxNum(max_expand)
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
    out = None

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
                "pools": {"recycling": bool(recycle), "states": int(caps["nodes"]), "predecessor_records": int(caps["edges"]), "open_log": int(caps["log"])},
                "helpers": {"per_leader": args.helpers, "reserved": args.help_reserved, **(pl.helperStats() if mine else {})},
                "parallelism": f"queries sharded, {world} map replica(s), RCCL broadcast",
            },
            "expansions_per_step": tot_exp,
            # what rank 0's searches did in the last step (the HBM traffic accounted by structure)
            "search_counters_rank0": {k: int(sum(getattr(r, k) for r in results)) for k in
                                      ("n_expanded", "n_nodes", "n_edges", "n_succ", "n_succ_finite", "voxel_reads", "n_push", "n_reopen", "n_refill", "n_evict")},
            # speculation accounting of rank 0's searches (mplx_result_speculation): units that ran get_succ vs units committed
            # (= n_expanded); the difference is work -- and traffic -- the K-way speculation threw away (batches cut ahead of a unit)
            "speculation": _speculation(pl, results),
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
        # HBM traffic of the same launch from the committed rocprofv3 PMC passes (tools/r06_final.sh; counters
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
            try:  # HBM traffic per streamed launch from the committed counter passes of the same leg (tools/r06_final.sh)
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
        tcaps = mapgen.c4_pools(jrk, max(len(tmine), 1), max_expand, per_q=args.max_nodes or (250_000 if (recycle and not jrk and len(tmine) >= 1024) else 0))
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
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    return out if rank == 0 else None


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
    recycle = os.environ.get("MPLX_BENCH_NO_RECYCLE") != "1"  # (the lanes take the planner's recycling policy: 250 000 states per query of the batch, see run())
    caps = mapgen.c4_pools(jrk, max(n_part, 256), max_expand, per_q=args.max_nodes or (((250_000 if recycle else 420_000) if split == 1 else 450_000) if not jrk else 0))
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


