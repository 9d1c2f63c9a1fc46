"""Latency probe (GPU box): per-expansion latency of single queries and the batch schedule."""
import sys, time, json
import numpy as np
sys.path.insert(0, '.')
from mpl_ros_amd import mapgen
from tests import util
from oracle import orc

which = sys.argv[1] if len(sys.argv) > 1 else 'single'
if which == 'single':
    grid, origin, res, start, goal, rng = mapgen.benchmark_map(256)
    U = mapgen.control_lattice(1.0, 1, True)
    mu, pl = util.make_gpu(grid, origin, res, U, v_max=2.0, a_max=1.0, max_nodes=1 << 21, max_edges=1 << 23, max_log=1 << 22)
    for it in range(2):
        ok = pl.plan(util.gpu_wp(start), util.gpu_wp(goal)); r = pl.getResult()
        print('C2 ACC', ok, r.cost, r.n_expanded, 'kernel ms', pl.lastKernelMs(), 'us/exp', 1e3 * pl.lastKernelMs() / r.n_expanded, 'refill', r.n_refill, 'evict', r.n_evict)
    U5 = mapgen.control_lattice(1.0, 2, True)
    mu, pl = util.make_gpu(grid, origin, res, U5, v_max=2.0, a_max=1.0, j_max=1.0, max_expand=20000, max_nodes=1 << 21, max_edges=1 << 23, max_log=1 << 22)
    ok = pl.plan(util.gpu_wp(start, control=orc.JRK), util.gpu_wp(goal, control=orc.JRK)); r = pl.getResult()
    print('C2 JRK cap20000', r.status, r.n_expanded, 'kernel ms', pl.lastKernelMs(), 'us/exp', 1e3 * pl.lastKernelMs() / r.n_expanded, 'refill', r.n_refill, 'evict', r.n_evict)
