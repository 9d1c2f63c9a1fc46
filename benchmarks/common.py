"""Shared pieces of the bench legs: progress log, the algorithmic-bytes formula of SURVEY.md 8(d), the CPU-baseline legs
(the oracle is the CHECKER and the reported baseline here -- never the thing measured)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
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


