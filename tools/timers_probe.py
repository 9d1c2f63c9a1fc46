"""Per-phase cycle breakdown of one long single query (C2 map) from the -DMPLX_LOOKUP_TIMERS build.
usage: MPLX_LIB=build_tmp/libmplx_timers.so python tools/timers_probe.py [helpers]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpl_ros_amd import mapgen
from tests import util
helpers = int(sys.argv[1]) if len(sys.argv) > 1 else -1
grid, origin, res, start, goal, rng = mapgen.benchmark_map(256)
U = mapgen.control_lattice(1.0, 1, True)
mu, pl = util.make_gpu(grid, origin, res, U, v_max=2.0, a_max=1.0, max_nodes=1 << 22, max_edges=1 << 24, max_log=1 << 23)
pl.setHelpers(helpers, -1)
for it in range(2):
    ok = pl.plan(util.gpu_wp(start), util.gpu_wp(goal)); r = pl.getResult()
    cy = pl.queryCycles()
    print('C2 ACC helpers', helpers, ok, r.n_expanded, 'kernel ms', pl.lastKernelMs(), 'us/exp', 1e3 * pl.lastKernelMs() / r.n_expanded,
          {k: round(v / r.n_expanded) for k, v in cy.items() if k not in ('batches', 'dep_batches', 'cache_hits')}, 'exp/batch', round(r.n_expanded / cy['batches'], 2),
          'hits', cy['cache_hits'], flush=True)
