"""One query of the C4 batch on its own, helpers from the start: per-phase cycles of the tail regime.  Query 1005 of the
stream is the one that runs into the 2 000 000 cap (12th in the longest-first launch order); 12 is a 388 k-expansion one.
With MPLX_LIB=build_tmp/libmplx_timers.so the kernel prints the fine-grained slots.  BIG=1: pools and state table sized
as for the whole 1024-query batch (the table then spans 8 GB instead of 256 MB).
usage: [BIG=1] [MPLX_LIB=...] python tools/tail_probe.py [query index] [helpers]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpl_ros_amd import mapgen
from tests import util
qi = int(sys.argv[1]) if len(sys.argv) > 1 else 12
helpers = int(sys.argv[2]) if len(sys.argv) > 2 else -1
grid, origin, res, start, goal, _ = mapgen.benchmark_map(512)
grid = np.ascontiguousarray(grid)
U = mapgen.control_lattice(1.0, 1, True)
queries = mapgen.c4_queries(grid, origin, res, 1024, rank=0)
s, g = queries[qi]
if os.environ.get("BIG"):
    pools = mapgen.c4_pools(False, 1024, 2_000_000)
    mu, pl = util.make_gpu(grid, origin, res, U, v_max=2.0, a_max=1.0, tol_pos=0.5, max_expand=2_000_000, n_slots=1024, max_nodes=pools["nodes"], max_edges=pools["edges"], max_log=pools["log"])
else:
    mu, pl = util.make_gpu(grid, origin, res, U, v_max=2.0, a_max=1.0, tol_pos=0.5, max_expand=2_000_000, max_nodes=1 << 24, max_edges=1 << 26, max_log=1 << 25)
pl.setHelpers(helpers, -1)
for it in range(2):
    ok = pl.plan(util.gpu_wp(s), util.gpu_wp(g)); r = pl.getResult()
    cy = pl.queryCycles()
    print(f"query {qi} helpers {helpers} it {it}: status {r.status} expansions {r.n_expanded} kernel {pl.lastKernelMs():.0f} ms = {1e3 * pl.lastKernelMs() / r.n_expanded:.3f} us/exp; "
          f"cycles/exp {({k: round(v / r.n_expanded) for k, v in cy.items() if k not in ('batches', 'dep_batches', 'cache_hits')})} exp/batch {r.n_expanded / cy['batches']:.2f} "
          f"dep batches {cy['dep_batches']} hits {cy['cache_hits']} {pl.helperStats()}", flush=True)
