"""One long single query (C2 map, ACC lattice) -- a clean target for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpl_ros_amd import mapgen
from tests import util
grid, origin, res, start, goal, rng = mapgen.benchmark_map(256)
U = mapgen.control_lattice(1.0, 1, True)
mu, pl = util.make_gpu(grid, origin, res, U, v_max=2.0, a_max=1.0, max_nodes=1 << 22, max_edges=1 << 24, max_log=1 << 23, spec=int(os.environ.get('SPEC', '-1')))
ok = pl.plan(util.gpu_wp(start), util.gpu_wp(goal)); r = pl.getResult()
print('C2 ACC', ok, r.cost, r.n_expanded, 'kernel ms', pl.lastKernelMs())
