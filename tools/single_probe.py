"""Single-query timings (GPU box): BASELINE C2 (256^3, 27-input ACC lattice, to the goal) and C3 (512^3, 125-input JRK
lattice, max_expand 2 000 000), with and without helper workgroups.  usage: python tools/single_probe.py [c2|c3|both]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpl_ros_amd import mapgen
from tests import util
which = sys.argv[1] if len(sys.argv) > 1 else "both"
if which in ("c2", "both"):
    grid, origin, res, start, goal, _ = mapgen.benchmark_map(256)
    U = mapgen.control_lattice(1.0, 1, True)
    for helpers in (0, -1):
        mu, pl = util.make_gpu(grid, origin, res, U, v_max=2.0, a_max=1.0, tol_pos=0.5, max_nodes=1 << 22, max_edges=1 << 24, max_log=1 << 23)
        pl.setHelpers(helpers, -1)
        for it in range(3):
            ok = pl.plan(util.gpu_wp(start), util.gpu_wp(goal)); r = pl.getResult()
        print(f"C2 helpers {helpers}: ok {ok} expansions {r.n_expanded} kernel {pl.lastKernelMs():.2f} ms ({r.n_expanded / pl.lastKernelMs() * 1e3:.0f} /s) "
              f"cache hits {pl.queryCycles()['cache_hits']} {pl.helperStats()}", flush=True)
        del mu, pl
if which in ("c3", "both"):
    grid, origin, res, start, goal, _ = mapgen.benchmark_map(512)
    U = mapgen.control_lattice(1.0, 2, True)
    for helpers in (0, -1):
        mu, pl = util.make_gpu(grid, origin, res, U, v_max=2.0, a_max=1.0, j_max=1.0, tol_pos=0.5, max_expand=2_000_000, max_nodes=1 << 26, max_edges=1 << 28, max_log=1 << 27)
        pl.setHelpers(helpers, -1)
        for it in range(2):
            ok = pl.plan(util.gpu_wp(start, control=7), util.gpu_wp(goal, control=7)); r = pl.getResult()
            print(f"C3 helpers {helpers} it {it}: status {r.status} expansions {r.n_expanded} kernel {pl.lastKernelMs():.0f} ms ({r.n_expanded / pl.lastKernelMs() * 1e3:.0f} /s) "
                  f"cache hits {pl.queryCycles()['cache_hits']} {pl.helperStats()}", flush=True)
        del mu, pl
