"""The C4-JRK batch at the survey's cap (SURVEY.md 8d C4 = "ACC and JRK variants" of the C3 set-up: 512^3, 125-input jerk lattice,
max_num 2 000 000) on ONE GPU: 1024 queries, each up to 2 M expansions and ~16 M states (C3: 16.0 M states, 51 M predecessor
records, 18 M OPEN-log entries).

What makes it fit (round 6): the pools are recycled (mplx_set_pool_recycling), so they hold the 36 queries that run at a time
(36 x 25 M = 900 M states, 144 GB) and not the batch's sum (17.6 G states).  What recycling does NOT shrink is the shared state table:
its slots are epoch-tagged per LAUNCH, so the entries of a launch's finished queries stay in it until the launch ends -- the table
(2^32 slots at most, 32 GB) has to hold the states a launch CREATES.  The batch therefore goes down in launches of 72 queries (two
rounds of the 36 leading workgroups, <= 1.5 G entries, load <= 0.35); the next launch's epoch makes the slots empty again without a
clear.  Every query's result is what it is in any other batching (queries are independent).

Parity sample: every 64-th query replayed on the CPU checker at the full cap, 16 worker processes (a 2 M-expansion jerk search
holds ~4 GB on the host).  Prints one JSON line.   usage (GPU box): python tools/c4jrk_full_cap.py [out.json] [queries per launch = 2 x leading workgroups]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks import common as bench
from mpl_ros_amd import mapgen
from mpl_ros_amd.planner import JRK, VoxelMapPlanner, VoxelMapUtil, Waypoint3D
from oracle import orc

NQ = int(os.environ.get("C4JRK_QUERIES", "1024"))
SLOTS, CAP = int(os.environ.get("C4JRK_SLOTS", "36")), 2_000_000
PER_LAUNCH = int(sys.argv[2]) if len(sys.argv) > 2 else 2 * SLOTS
n, res, origin = 512, 0.1, (0.0, 0.0, 0.0)
grid, _, _, _, _, _ = mapgen.benchmark_map(n)
queries = mapgen.c4_queries(grid, origin, res, 1024, rank=0)[:NQ]
U = mapgen.control_lattice(1.0, 2, True)
mu = VoxelMapUtil(0)
mu.setMap(origin, (n, n, n), grid.ravel(), res)
pl = VoxelMapPlanner(False)
pl.setMapUtil(mu); pl.setVmax(2.0); pl.setAmax(1.0); pl.setJmax(1.0); pl.setDt(1.0); pl.setU(U); pl.setTol(0.5); pl.setMaxNum(CAP)
POOL = dict(nodes=SLOTS * 25_000_000, edges=SLOTS * 80_000_000, log=SLOTS * 34_000_000)  # (per query: C3 makes 16.0 M / 51 M / 18 M; the batch's capped queries ~20.5 M states;
# measured: 48 x 17 M left 146 queries reporting POOL_FULL, 40 x 22 M 15)
pl.setCapacity(SLOTS, POOL["nodes"], POOL["edges"], POOL["log"])
pl.setPoolRecycling(True)
pl.setDeadline(float(os.environ.get("MPLX_DEADLINE_S", "300")))


def wp(p):
    w = Waypoint3D(JRK)
    w.pos = np.array(p, dtype=np.float64)
    return w


def word(r):
    return (r.status, r.traj_len, r.cost, r.n_expanded, r.n_nodes, r.n_edges, r.n_succ_finite, r.voxel_reads, r.expand_hash)


sample = list(range(7, NQ, 64))
R, traj, launches = [], {}, []
t0 = time.perf_counter()
for a in range(0, NQ, PER_LAUNCH):
    part = queries[a:a + PER_LAUNCH]
    t1 = time.perf_counter()
    rr = pl.planBatch([wp(s) for s, g in part], [wp(g) for s, g in part])
    launches.append({"queries": len(part), "kernel_ms": round(pl.lastKernelMs(), 1), "wall_ms": round(1e3 * (time.perf_counter() - t1), 1),
                     "expansions": int(sum(r.n_expanded for r in rr)), "states_created": int(sum(r.n_nodes for r in rr))})
    for k in sample:
        if a <= k < a + len(part):
            traj[k] = pl.getTraj(k - a).actions.copy()
    for k, r in enumerate(rr):  # a query that reports POOL_FULL: which of its three resources is at a limit
        if r.status == 4:
            print(f"  query {a + k} POOL_FULL: expansions {r.n_expanded} states {r.n_nodes} predecessor records {r.n_edges} OPEN-log entries {r.n_push}", file=sys.stderr, flush=True)
    R += rr
    print(f"launch {len(launches)}: {launches[-1]}", file=sys.stderr, flush=True)
gpu_s = time.perf_counter() - t0
n_exp = int(sum(r.n_expanded for r in R))
status = np.bincount(np.array([r.status for r in R], dtype=np.int64), minlength=7)[:7].tolist()
# the same queries planned one at a time (mplx_plan: no recycling, no batch) must give the same words: two of the sample
single = {}
pl.setPoolRecycling(False)
for k in sample[:2]:
    s, g = queries[k]
    pl.plan(wp(s), wp(g))
    single[k] = word(pl.getResult()) == word(R[k])
del pl, mu
# CPU replay of the sample at the full cap
native = orc.use_native()
map_path = f"/dev/shm/mplx_c4jrk_cap_{os.getpid()}.npy"
np.save(map_path, np.ascontiguousarray(grid, dtype=np.int8))
kw = dict(dt=1.0, v_max=2.0, a_max=1.0, j_max=1.0, tol_pos=0.5, max_expand=CAP)
cfg = {"map": map_path, "origin": list(origin), "res": res, "control": int(orc.JRK), "U": U.tolist(), "kw": kw, "native": bool(native)}
try:
    t1 = time.time()
    st = bench._cpu_run(cfg, queries, sample, 3000.0, min(os.cpu_count() or 1, 16, len(sample)))
    cpu_s = time.time() - t1
finally:
    os.remove(map_path)
bad, not_comparable = [], []
for k, (ne, nn, cost, h, actions) in st["per_query"].items():
    r = R[k]
    if r.status == 4:  # the device reported this query's pools full (never silently truncated): nothing to compare
        not_comparable.append(k)
        continue
    ok = ne == r.n_expanded and nn == r.n_nodes and h == r.expand_hash and (cost == r.cost or (np.isinf(r.cost) and not np.isfinite(cost)))
    if ok and actions is not None:
        ok = np.array_equal(traj[k], actions)
    if not ok:
        bad.append(k)
out = {"metric": "node_expansions_per_s", "value": n_exp / gpu_s, "unit": "expansions/s", "n_gpus": 1, "seconds": round(gpu_s, 2),
       "config": {"workload": f"C4-JRK at the survey's cap: {NQ} queries, 512^3 random-box map, 125-input jerk lattice, max_expand {CAP}",
                  "launches": len(launches), "queries_per_launch": PER_LAUNCH, "leading_workgroups": SLOTS,
                  "pools": {"recycling": True, "states": POOL["nodes"], "predecessor_records": POOL["edges"], "open_log": POOL["log"]}},
       "expansions": n_exp, "states_created": int(sum(r.n_nodes for r in R)),
       "plan_status_counts": {"ok": status[0], "no_path": status[1], "start_occupied": status[2], "max_expand": status[3], "pool_full": status[4],
                              "internal": status[5], "traj_too_long": status[6]},
       "launch_table": launches,
       "parity_sample": {"queries": sorted(st["per_query"].keys()), "mismatches": len(bad), "first_bad": bad[:5], "reported_pool_full_on_device": not_comparable,
                         "device_status": {int(k): int(R[k].status) for k in sorted(st["per_query"].keys())},
                         "checked": "n_expanded, n_nodes, expand_hash (order-dependent), cost (bit-exact f64), path actions -- CPU checker at the full cap",
                         "cpu_wall_s": round(cpu_s, 1), "cpu_expansions_per_s_per_core": st["n_exp"] / max(st["busy"], 1e-9) if st.get("busy") else None,
                         "same_words_from_single_plans_without_recycling": single}}
print(json.dumps(out), flush=True)
if len(sys.argv) > 1 and sys.argv[1] != "-":
    json.dump(out, open(sys.argv[1], "w"), indent=1)
