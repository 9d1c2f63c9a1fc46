"""Helper-workgroup probe (GPU box): a C4-like batch at reduced size under several helper settings."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpl_ros_amd import mapgen
from tests import util

n = int(os.environ.get("MAP", "256")); nq = int(os.environ.get("NQ", "300")); cap = int(os.environ.get("CAP", "30000"))
per, res_ = int(sys.argv[1]), int(sys.argv[2])
grid, origin, res, start, goal, rng = mapgen.benchmark_map(n)
U = mapgen.control_lattice(1.0, 1, True)
queries = mapgen.c4_queries(grid, origin, res, nq, 0, min_dist=8.0)
mu, pl = util.make_gpu(grid, origin, res, U, v_max=2.0, a_max=1.0, n_slots=1024, max_expand=cap, max_nodes=cap * 4 * nq, max_edges=cap * 16 * nq, max_log=cap * 6 * nq)
pl.setHelpers(per, res_)
starts = [util.gpu_wp(s) for s, g in queries]; goals = [util.gpu_wp(g) for s, g in queries]
iters = int(os.environ.get("ITER", "3"))
times = []
for it in range(iters):
    t = time.time(); R = pl.planBatch(starts, goals); wall = time.time() - t
    times.append(pl.lastKernelMs())
    ne = sum(r.n_expanded for r in R)
    st = pl.helperStats()
    if iters <= 3 or times[-1] > 3 * np.median(times) or st["helpers_gave_up"]:
        hits = sum(pl.queryCycles(k)["cache_hits"] for k in range(nq))
        T = np.array([pl.queryTiming(k) for k in range(nq)])
        late = np.argsort(-T[:, 1])[:3]
        print(f"helpers {per} reserved {res_} it {it}: wall {wall:.3f} s kernel {times[-1]:.1f} ms expansions {ne} cache hits {hits} {st} "
              f"latest queries {[(int(k), round(T[k,0],3), round(T[k,1],3), int(R[k].n_expanded), int(R[k].status)) for k in late]}", flush=True)
print(f"helpers {per} reserved {res_}: {iters} steps, kernel ms median {np.median(times):.1f} max {np.max(times):.1f}", flush=True)
