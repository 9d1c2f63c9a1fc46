import sys, os
sys.path.insert(0, '.')
import numpy as np
from mpl_ros_amd import mapgen
from tests import util
from oracle import orc
spec = int(sys.argv[1]); me = int(sys.argv[2]); hid = bool(int(sys.argv[3]))
grid, origin, res, start, goal, rng = mapgen.benchmark_map(256)
U5 = mapgen.control_lattice(1.0, 2, True)
mu, pl = util.make_gpu(grid, origin, res, U5, v_max=2.0, a_max=1.0, j_max=1.0, max_expand=me, max_nodes=1 << 22, max_edges=1 << 24, max_log=1 << 23, spec=spec, heur_ignore_dynamics=hid)
ok = pl.plan(util.gpu_wp(start, control=orc.JRK), util.gpu_wp(goal, control=orc.JRK)); r = pl.getResult()
print('JRK', sys.argv[1:], r.status, r.n_expanded, r.n_nodes, 'ms', pl.lastKernelMs(), 'refill', r.n_refill, 'evict', r.n_evict)
