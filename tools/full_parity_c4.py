"""Every one of the 1024 queries of the C4-ACC batch (bench.py's default workload, cap 2 000 000) replayed on the CPU checker
(64 worker processes, longest first) and compared with the GPU batch: status, expansions, states created, expansion-order
hash, cost (bit-exact f64), path actions.  ~4 minutes of host time.  usage (GPU box): python tools/full_parity_c4.py [out.json] [acc|jrk]
(jrk: the 125-input jerk lattice, cap 20 000 per query = bench.py --lattice jrk)"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks import common as bench
from mpl_ros_amd import mapgen
from mpl_ros_amd.planner import ACC, JRK, VoxelMapPlanner, VoxelMapUtil, Waypoint3D
from oracle import orc

jrk = len(sys.argv) > 2 and sys.argv[2] == "jrk"
CONTROL = JRK if jrk else ACC
n, res, origin, cap = 512, 0.1, (0.0, 0.0, 0.0), (20000 if jrk else 2_000_000)
grid, _, _, _, _, _ = mapgen.benchmark_map(n)
queries = mapgen.c4_queries(grid, origin, res, 1024, rank=0)
U = mapgen.control_lattice(1.0, 2 if jrk else 1, True)
mu = VoxelMapUtil(0)
mu.setMap(origin, (n, n, n), grid.ravel(), res)
pl = VoxelMapPlanner(False)
pl.setMapUtil(mu); pl.setVmax(2.0); pl.setAmax(1.0); pl.setDt(1.0); pl.setU(U); pl.setTol(0.5); pl.setMaxNum(cap)
if jrk:
    pl.setJmax(1.0)
caps = mapgen.c4_pools(jrk, 1024, cap)
pl.setCapacity(768 if jrk else 1024, caps["nodes"], caps["edges"], caps["log"])


def wp(p):
    w = Waypoint3D(CONTROL)
    w.pos = np.array(p, dtype=np.float64)
    return w


t0 = time.time()
R = pl.planBatch([wp(s) for s, g in queries], [wp(g) for s, g in queries])
print(f"GPU batch: {sum(r.n_expanded for r in R)} expansions, kernel {pl.lastKernelMs():.0f} ms", flush=True)
gpu_exp = [r.n_expanded for r in R]
native = orc.use_native()
map_path = f"/dev/shm/mplx_full_parity_{os.getpid()}.npy"
np.save(map_path, np.ascontiguousarray(grid, dtype=np.int8))
kw = dict(dt=1.0, v_max=2.0, a_max=1.0, tol_pos=0.5, max_expand=cap)
if jrk:
    kw["j_max"] = 1.0
cfg = {"map": map_path, "origin": list(origin), "res": res, "control": int(orc.JRK if jrk else orc.ACC), "U": U.tolist(), "kw": kw, "native": bool(native)}
order = sorted(range(1024), key=lambda i: -gpu_exp[i])
try:
    t1 = time.time()
    st = bench._cpu_run(cfg, queries, order, 3000.0, min(os.cpu_count() or 1, 64))
    cpu_s = time.time() - t1
finally:
    os.remove(map_path)
bad = []
for k, (ne, nn, cost, h, actions) in st["per_query"].items():
    r = R[k]
    ok = ne == r.n_expanded and nn == r.n_nodes and h == r.expand_hash and (cost == r.cost or (np.isinf(r.cost) and not np.isfinite(cost)))
    if ok and actions is not None:
        ok = np.array_equal(pl.getTraj(k).actions, actions)
    if not ok:
        bad.append(k)
out = {"workload": f"C4-{'JRK' if jrk else 'ACC'}, 1024 queries, 512^3, {U.shape[0]}-input lattice, max_expand {cap} (bench.py{' --lattice jrk' if jrk else ' default'})", "queries_replayed_on_cpu": len(st["per_query"]), "mismatches": len(bad),
       "first_bad": bad[:5], "expansions": int(sum(gpu_exp)), "cpu_wall_s": round(cpu_s, 1), "cpu_processes": min(os.cpu_count() or 1, 64),
       "cpu_expansions_per_s": st["n_exp"] / max(st["wall"], 1e-9),
       "checked": "status-independent: n_expanded, n_nodes, expand_hash (order-dependent), cost (bit-exact f64), path actions"}
print(json.dumps(out), flush=True)
if len(sys.argv) > 1 and sys.argv[1] != "-":
    json.dump(out, open(sys.argv[1], "w"), indent=1)
