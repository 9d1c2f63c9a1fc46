import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpl_ros_amd import mapgen
from tests import util
grid, origin, res, start, goal, rng = mapgen.benchmark_map(256)
U = mapgen.control_lattice(1.0, 1, True)
mu, pl = util.make_gpu(grid, origin, res, U, v_max=2.0, a_max=1.0, max_nodes=1 << 22, max_edges=1 << 24, max_log=1 << 23)
ok = pl.plan(util.gpu_wp(start), util.gpu_wp(goal)); r = pl.getResult()
ctx = pl._ctx(); cyc = (C.c_uint64 * 10)(); ctx.lib.mplx_result_cycles(ctx.h, 0, cyc)
nb = cyc[7]
print('per batch:', {n: round(cyc[i] / nb) for i, n in enumerate(['pop', 'expand', 'lookup', 'P1(thread0)', 'refill', 'scan+owner+barrier', 'ordered', '-', '-', 'P2(thread0)'])}, 'batches', nb, 'ms', pl.lastKernelMs())
