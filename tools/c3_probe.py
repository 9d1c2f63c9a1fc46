"""C3 (512^3, 125-input JRK lattice) with an expansion cap, helpers on: per-phase cycles; with MPLX_LIB=build_tmp/libmplx_timers.so
the kernel prints the fine-grained slots of the batch loop.  usage: [MPLX_LIB=...] python tools/c3_probe.py [cap]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpl_ros_amd import mapgen
from tests import util
cap = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
grid, origin, res, start, goal, _ = mapgen.benchmark_map(512)
U = mapgen.control_lattice(1.0, 2, True)
mu, pl = util.make_gpu(grid, origin, res, U, v_max=2.0, a_max=1.0, j_max=1.0, tol_pos=0.5, max_expand=cap, max_nodes=1 << 25, max_edges=1 << 27, max_log=1 << 26)
for it in range(2):
    ok = pl.plan(util.gpu_wp(start, control=7), util.gpu_wp(goal, control=7)); r = pl.getResult()
    cy = pl.queryCycles()
    print(f"C3 cap {cap} it {it}: status {r.status} expansions {r.n_expanded} kernel {pl.lastKernelMs():.0f} ms = {1e3 * pl.lastKernelMs() / r.n_expanded:.3f} us/exp; "
          f"cycles/exp {({k: round(v / r.n_expanded) for k, v in cy.items() if k not in ('batches', 'dep_batches', 'cache_hits')})} exp/batch {r.n_expanded / cy['batches']:.2f} "
          f"dep batches {cy['dep_batches']} hits {cy['cache_hits']} succ/exp {r.n_succ / r.n_expanded:.1f} finite/exp {r.n_succ_finite / r.n_expanded:.1f} new/exp {r.n_nodes / r.n_expanded:.2f} {pl.helperStats()}", flush=True)
