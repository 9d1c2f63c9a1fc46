"""Fixed cost of a search launch: the skir query (333 expansions) and a query that ends at once (start inside the goal
region), default kernel / helpers off / one-node kernel.  usage: python tools/fixed_cost_probe.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpl_ros_amd import mapgen
from tests import util
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "skir_map.npz"))
grid, origin, res = d["grid"], d["origin"], float(d["res"])
U = mapgen.control_lattice(1.0, 1, True)
for name, helpers, spec in (("default (spec + helpers)", -1, -1), ("spec, helpers off", 0, -1), ("one-node kernel", 0, 0)):
    mu, pl = util.make_gpu(grid, origin, res, U, v_max=2.0, a_max=1.0, tol_pos=0.5, spec=spec)
    pl.setHelpers(helpers, -1)
    for label, s, g in (("skir query", ((5.5, 5.5, 0.5), (1, 0, 0)), (1.5, 1.5, 5.5)), ("start in goal region", ((5.5, 5.5, 0.5), (0, 0, 0)), (5.6, 5.5, 0.5))):
        ts = []
        for it in range(5):
            pl.plan(util.gpu_wp(s[0], vel=s[1]), util.gpu_wp(g))
            ts.append(pl.lastKernelMs())
        print(f"{name:28s} {label:22s} expansions {pl.getResult().n_expanded:4d} kernel ms {['%.3f' % t for t in ts]} {pl.kernelName()}", flush=True)
mu, pl = util.make_gpu(grid, origin, res, U, v_max=2.0, a_max=1.0, tol_pos=0.5, spec=0)
pl.setHelpers(0, -1)
pl.plan(util.gpu_wp((5.5, 5.5, 0.5), vel=(1, 0, 0)), util.gpu_wp((1.5, 1.5, 5.5)))
r = pl.getResult()
print("one-node kernel cycles per expansion:", {k: round(v / max(r.n_expanded, 1)) for k, v in pl.queryCycles().items()}, "nodes", r.n_nodes, "succ", r.n_succ, "finite", r.n_succ_finite, "reads", r.voxel_reads, "push", r.n_push, "refill", r.n_refill, "evict", r.n_evict)
