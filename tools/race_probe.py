"""The same query planned again and again: every result word must repeat exactly (a race between waves, or between a leader
and its helpers, shows as a different expansion-order hash).  usage: python tools/race_probe.py c3|acc [runs] [helpers]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpl_ros_amd import mapgen
from tests import util
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
helpers = int(sys.argv[3]) if len(sys.argv) > 3 else -1
grid, origin, res, start, goal, _ = mapgen.benchmark_map(512)
grid = np.ascontiguousarray(grid)
if which == "c3":
    U = mapgen.control_lattice(1.0, 2, True)
    mu, pl = util.make_gpu(grid, origin, res, U, v_max=2.0, a_max=1.0, j_max=1.0, tol_pos=0.5, max_expand=2_000_000, max_nodes=1 << 26, max_edges=1 << 28, max_log=1 << 27)
    s, g, ctl = start, goal, 7
else:
    U = mapgen.control_lattice(1.0, 1, True)
    queries = mapgen.c4_queries(grid, origin, res, 1024, rank=0)
    s, g = queries[1005]
    ctl = 3
    mu, pl = util.make_gpu(grid, origin, res, U, v_max=2.0, a_max=1.0, tol_pos=0.5, max_expand=2_000_000, max_nodes=1 << 24, max_edges=1 << 26, max_log=1 << 25)
pl.setHelpers(helpers, -1)
seen = {}
for it in range(runs):
    pl.plan(util.gpu_wp(s, control=ctl) if which == "c3" else util.gpu_wp(s), util.gpu_wp(g, control=ctl) if which == "c3" else util.gpu_wp(g))
    r = pl.getResult()
    key = (r.status, r.n_expanded, r.expand_hash, r.n_nodes, r.n_edges, r.voxel_reads, r.n_succ, r.n_succ_finite)
    seen[key] = seen.get(key, 0) + 1
    print(f"{which} run {it}: hash {r.expand_hash:016x} nodes {r.n_nodes} edges {r.n_edges} reads {r.voxel_reads} kernel {pl.lastKernelMs():.0f} ms", flush=True)
print("DISTINCT RESULTS:", len(seen), "(must be 1)")
