#!/usr/bin/env python3
"""bench.py -- node-expansions/s of the device-resident A* on the BASELINE.json C4 workload.

One "step" = one pass of the hot path over one batch of queries: `mplx_plan_batch` of Q independent
start/goal queries on the shared 512^3 random-box voxel map (map generator of C3).
Inputs (map replica, control set) are resident in HBM before the timed region; queries are a few
hundred bytes each.  Weak scaling: every rank plans its own Q queries (rank-specific PRNG stream) on
its own replica of the map, which rank 0 generates and RCCL-broadcasts over xGMI; no collective on
the data path.

    python bench.py                       # 1 GPU, defaults
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is the astar kernel: algorithmic bytes (SURVEY.md 8d
B_exp, from the kernel's own counters) / its HIP-event duration; `cpu_baseline` is the CPU oracle
(oracle/, "port") timed on a bounded sample of the same queries on one host core.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

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
    ap.add_argument("--queries", type=int, default=1024, help="queries per GPU per step")
    ap.add_argument("--map", type=int, default=512, help="voxel map edge length")
    ap.add_argument("--lattice", choices=["acc", "jrk"], default="acc")
    ap.add_argument("--single", action="store_true",
                    help="BASELINE.md C3: ONE query (2.05,2.05,2.05)->(49.15,49.15,49.15) on the 512^3 map instead of the C4 batch")
    ap.add_argument("--max-expand", type=int, default=0,
                    help="per-query expansion cap setMaxNum (default: 2 000 000 for acc = the BASELINE.md C3 cap, 20000 for jrk)")
    ap.add_argument("--slots", type=int, default=0)
    ap.add_argument("--max-nodes", type=int, default=0, help="mean states per query used to size the shared pools")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline sample budget (0 disables)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="host threads of the CPU baseline (0 = all cores, at most 64; a single query uses 1)")
    args = ap.parse_args()
    if args.single:
        args.queries = 1

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    from mpl_ros_amd import mapgen
    from mpl_ros_amd.planner import ACC, JRK, VoxelMapPlanner, VoxelMapUtil, Waypoint3D

    control = ACC if args.lattice == "acc" else JRK
    n = args.map
    res = 0.1
    origin = (0.0, 0.0, 0.0)

    # ---- map: rank 0 generates, RCCL broadcast puts one replica in every GPU's HBM
    t0 = time.time()
    if rank == 0:
        grid, _, _, _, _, _ = mapgen.benchmark_map(n)
        map_t = torch.from_numpy(grid.reshape(-1)).to(dev)
    else:
        grid = None
        map_t = torch.empty(n * n * n, dtype=torch.int8, device=dev)
    t_gen = time.time() - t0
    t0 = time.time()
    if world > 1:
        dist.broadcast(map_t, src=0)
        if grid is None:
            grid = map_t.cpu().numpy().reshape(n, n, n)  # host copy only to draw free query cells
    torch.cuda.synchronize()
    t_bcast = time.time() - t0

    mu = VoxelMapUtil(local_rank)
    mu.setMapDevice(map_t.data_ptr(), origin, (n, n, n), res)

    # ---- planner: C4 parameters
    if control == ACC:
        U = mapgen.control_lattice(1.0, 1, True)
        # BASELINE.md bounds wall time with max_num = 2 000 000 expansions (C3); the same cap is applied to
        # the C4 queries: one of the 1024 random pairs has a goal that is not reachable within it
        max_expand = args.max_expand if args.max_expand > 0 else 2_000_000
        per_q = args.max_nodes or 450_000  # mean states per query (tail up to ~2 M; the pools are shared)
        caps = dict(nodes=per_q * args.queries, edges=per_q * args.queries * 9 // 2, log=per_q * args.queries * 5 // 4)
        slots = args.slots or 1024
    else:
        U = mapgen.control_lattice(1.0, 2, True)
        max_expand = args.max_expand if args.max_expand > 0 else (2_000_000 if args.single else 20000)
        # the 125-primitive lattice creates ~20 states and ~50 predecessor records per expansion
        per_q = args.max_nodes or max(1 << 16, max_expand * 24)
        caps = dict(nodes=per_q * args.queries, edges=per_q * args.queries * 7 // 2, log=per_q * args.queries * 3 // 2)
        slots = args.slots or 768
    pl = VoxelMapPlanner(False)
    pl.setMapUtil(mu)
    pl.setVmax(2.0)
    pl.setAmax(1.0)
    if control == JRK:
        pl.setJmax(1.0)
    pl.setDt(1.0)
    pl.setU(U)
    pl.setTol(0.5)
    pl.setMaxNum(max_expand)
    pl.setCapacity(min(slots, args.queries), caps["nodes"], caps["edges"], caps["log"])

    # ---- queries: rank-specific stream, free cell centres >= 10 m apart
    qrng = mapgen.SplitMix64(20250620 + 7919 * (rank + 1))
    if args.single:
        g = {256: 23.55, 512: 49.15}.get(n, round((n - 20) * res, 2) + 0.05)
        queries = [((2.05, 2.05, 2.05), (g, g, g))]  # the bubbles carved by benchmark_map()
        args.queries = 1
    else:
        queries = mapgen.random_queries(grid, origin, res, args.queries, qrng, min_dist=10.0)

    def wp(p):
        w = Waypoint3D(control)
        w.pos = np.array(p, dtype=np.float64)
        return w

    starts = [wp(s) for s, g in queries]
    goals = [wp(g) for s, g in queries]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        pl.planBatch(starts, goals)
    barrier()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    results = None
    for _ in range(args.steps):
        results = pl.planBatch(starts, goals)
        kernel_ms += pl.lastKernelMs()
    barrier()
    elapsed = time.perf_counter() - t0

    n_exp = sum(r.n_expanded for r in results)
    reads = sum(r.voxel_reads for r in results)
    nsf = sum(r.n_succ_finite for r in results)
    status = np.bincount([r.status for r in results], minlength=5)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([n_exp, reads, nsf] + status.tolist(), dtype=torch.int64, device=dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        tot_exp = int(c[0].item())
        tot_status = c[3:].tolist()
    else:
        tot_exp = n_exp
        tot_status = status.tolist()

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = tot_exp * args.steps / elapsed
        k_ms = kernel_ms / args.steps  # rank 0's astar kernel, HIP events on its launch stream
        alg = algorithmic_bytes(control, n_exp, reads, nsf)
        achieved = alg / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "node_expansions_per_s",
            "value": value,
            "unit": "expansions/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (f"C3-{args.lattice.upper()}: single query (2.05,..)->({queries[0][1][0]},..) on a " if args.single else
                             f"C4-{args.lattice.upper()}: {args.queries} independent start/goal queries per GPU on one shared ") +
                            f"{n}^3 random-box voxel map (10% occupied, seed 20250620), {U.shape[0]}-primitive {args.lattice} lattice, "
                            f"dt 1 v_max 2 a_max 1 tol 0.5" + (f", max_expand {max_expand}" if max_expand > 0 else ""),
                "queries_per_gpu": args.queries,
                "map_dim": [n, n, n],
                "n_primitives": int(U.shape[0]),
                "slots_per_gpu": min(slots, args.queries),
                "parallelism": f"queries sharded, {world} map replica(s), RCCL broadcast",
            },
            "expansions_per_step": tot_exp,
            "plan_status_counts": {"ok": tot_status[0], "no_path": tot_status[1], "start_occupied": tot_status[2],
                                   "max_expand": tot_status[3], "pool_full": tot_status[4]},
            "plan_ms_mean_per_query": ms_per_step / args.queries,
            "map_setup_s": {"generate": round(t_gen, 3), "broadcast": round(t_bcast, 3)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "astar_spec_kernel<32,16,ACC>" if control == ACC else "astar_spec_kernel<128,4,JRK>", "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg,
                         "bytes_per_expansion": alg / max(n_exp, 1)},
        }
        # HBM traffic of the same launch from the committed rocprofv3 PMC passes (cannot be collected
        # inside this process); only attached when the profile is of this workload
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if args.lattice == "acc" and args.queries == 1024 and n == 512 and world == 1:
                out["roofline"]["traffic"] = (tr["FETCH_SIZE_KB"] + tr["WRITE_SIZE_KB"]) * 1024.0
                out["roofline"]["traffic_source"] = tr["profile"]
        except Exception:
            pass
        if args.cpu_seconds > 0 and world == 1:
            # a single capped query is sampled on the CPU with a smaller cap (same search, stopped earlier)
            cpu_cap = min(max_expand, 250_000) if (args.single and max_expand > 0) else max_expand
            nthr = 1 if args.single else (args.cpu_threads if args.cpu_threads > 0 else min(os.cpu_count() or 1, 64))
            out["cpu_baseline"] = cpu_baseline(grid, origin, res, control, U, cpu_cap, queries, args.cpu_seconds, nthr)
            # the CPU sample doubles as a full-size parity check of the timed GPU results (checker only):
            # expansions, states created and path cost of every sampled query must be identical
            pq = out["cpu_baseline"].pop("_per_query")
            if cpu_cap == max_expand:
                bad = [i for i, (ne, nn, cost) in pq.items()
                       if ne != results[i].n_expanded or nn != results[i].n_nodes
                       or not (cost == results[i].cost or (np.isinf(results[i].cost) and not np.isfinite(cost)))]
                out["parity_sample"] = {"queries": len(pq), "mismatches": len(bad), "checked": "n_expanded, n_nodes, cost (bit-exact f64)"}
                if bad:
                    out["parity_sample"]["first_bad_query"] = int(bad[0])
            if cpu_cap != max_expand:
                out["cpu_baseline"]["sample"] += f"; CPU run capped at {cpu_cap} expansions"
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(grid, origin, res, control, U, max_expand, queries, budget_s, threads=1):
    """The CPU oracle (oracle/, a restatement -- kind "port") on a bounded sample of the same
    queries: `threads` host threads, one query at a time each (the reference planner is
    single-threaded; independent queries are the only parallelism it offers), steady clock around
    the plan() calls only.  value = expansions of the sample / wall time of the sample."""
    import threading
    from oracle import orc
    kw = dict(dt=1.0, v_max=2.0, a_max=1.0, tol_pos=0.5, max_expand=max_expand)
    if control == orc.JRK:
        kw["j_max"] = 1.0
    threads = max(1, min(threads, len(queries)))
    planners = []
    for _ in range(threads):  # ctypes releases the GIL inside plan(): the threads run in parallel
        P = orc.Planner()
        P.set_map(grid, origin, res)
        P.set_config(control, U, **kw)
        planners.append(P)
    lock = threading.Lock()
    state = {"next": 0, "n_exp": 0, "nq": 0, "busy": 0.0, "per_query": {}}
    t_start = time.perf_counter()

    def work(P):
        while True:
            with lock:
                i = state["next"]
                if i >= len(queries) or time.perf_counter() - t_start >= budget_s:
                    return
                state["next"] = i + 1
            s, g = queries[i]
            P.reset_counters()
            t0 = time.perf_counter()
            P.plan(orc.waypoint(s, control=control), orc.waypoint(g, control=control))
            dt = time.perf_counter() - t0
            with lock:
                ne = P.counters()["n_expansions"]
                state["n_exp"] += ne
                state["nq"] += 1
                state["busy"] += dt
                state["per_query"][i] = (ne, P.num_nodes(), P.traj_cost)

    ths = [threading.Thread(target=work, args=(P,)) for P in planners]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    wall = time.perf_counter() - t_start
    n_exp, nq = state["n_exp"], state["nq"]
    return {"value": n_exp / wall, "unit": "expansions/s", "cores": threads, "kind": "port",
            "value_per_core": n_exp / state["busy"],
            "sample": f"first {nq} of the {len(queries)} queries of rank 0 ({n_exp} expansions, {wall:.1f} s wall, {state['busy']:.1f} core-s of plan())",
            "plan_ms_mean_per_query": 1e3 * state["busy"] / nq,
            "_per_query": state["per_query"]}


if __name__ == "__main__":
    main()
