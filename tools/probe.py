"""Latency probe (GPU box): per-expansion latency of single queries and the batch schedule."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpl_ros_amd import mapgen
from tests import util
from oracle import orc

which = sys.argv[1] if len(sys.argv) > 1 else 'single'
import os
SPEC = int(os.environ.get('SPEC', '-1'))
HID = bool(int(os.environ.get('HID', '0')))
if which == 'single':
    grid, origin, res, start, goal, rng = mapgen.benchmark_map(256)
    U = mapgen.control_lattice(1.0, 1, True)
    BIG = int(os.environ.get('BIG', '0'))
    mn = 460_000_000 if BIG else 1 << 22
    mu, pl = util.make_gpu(grid, origin, res, U, v_max=2.0, a_max=1.0, max_nodes=mn, max_edges=(mn * 9 // 2) if BIG else 1 << 24, max_log=(mn * 5 // 4) if BIG else 1 << 23, spec=SPEC, heur_ignore_dynamics=HID)
    for it in range(2):
        ok = pl.plan(util.gpu_wp(start), util.gpu_wp(goal)); r = pl.getResult()
        print('C2 ACC', ok, r.cost, r.n_expanded, 'kernel ms', pl.lastKernelMs(), 'us/exp', 1e3 * pl.lastKernelMs() / r.n_expanded, 'refill', r.n_refill, 'evict', r.n_evict)
        cy = pl.queryCycles(); print('   cycles/exp', {k: round(v / r.n_expanded) for k, v in cy.items()}, 'exp/batch', round(r.n_expanded / max(cy.get('batches', 0), 1), 3), 'dep frac', round(cy.get('dep_batches', 0) / max(cy.get('batches', 0), 1), 3))
    if BIG: sys.exit(0)
    U5 = mapgen.control_lattice(1.0, 2, True)
    mu, pl = util.make_gpu(grid, origin, res, U5, v_max=2.0, a_max=1.0, j_max=1.0, max_expand=20000, max_nodes=1 << 21, max_edges=1 << 23, max_log=1 << 22, spec=SPEC)
    ok = pl.plan(util.gpu_wp(start, control=orc.JRK), util.gpu_wp(goal, control=orc.JRK)); r = pl.getResult()
    print('C2 JRK cap20000', r.status, r.n_expanded, 'kernel ms', pl.lastKernelMs(), 'us/exp', 1e3 * pl.lastKernelMs() / r.n_expanded, 'refill', r.n_refill, 'evict', r.n_evict)
    cy = pl.queryCycles(); print('   cycles/exp', {k: round(v / r.n_expanded) for k, v in cy.items()}, 'exp/batch', round(r.n_expanded / max(cy.get('batches', 0), 1), 3), 'dep frac', round(cy.get('dep_batches', 0) / max(cy.get('batches', 0), 1), 3))
if which == 'batch':
    nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    slots = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    grid, origin, res, start, goal, rng = mapgen.benchmark_map(512)
    U = mapgen.control_lattice(1.0, 1, True)
    qrng = mapgen.SplitMix64(20250620 + 7919)
    queries = mapgen.random_queries(grid, origin, res, nq, qrng, min_dist=10.0)
    per_q = 450_000
    mu, pl = util.make_gpu(grid, origin, res, U, v_max=2.0, a_max=1.0, n_slots=slots, max_expand=int(os.environ.get('CAP', '-1')), max_nodes=per_q * nq, max_edges=per_q * nq * 9 // 2, max_log=per_q * nq * 5 // 4, spec=SPEC)
    starts = [util.gpu_wp(s) for s, g in queries]; goals = [util.gpu_wp(g) for s, g in queries]
    t = time.time(); R = pl.planBatch(starts, goals); wall = time.time() - t
    ne = np.array([r.n_expanded for r in R], dtype=np.float64)
    st = np.bincount([r.status for r in R], minlength=5)
    T = np.array([pl.queryTiming(q) for q in range(nq)])
    dur = T[:, 1] - T[:, 0]
    print('wall', wall, 'kernel ms', pl.lastKernelMs(), 'status', st, 'expansions', ne.sum(), 'exp/s', ne.sum() / (pl.lastKernelMs() * 1e-3))
    print('makespan', T[:, 1].max(), 'sum dur', dur.sum(), 'utilisation', dur.sum() / (min(slots, nq) * T[:, 1].max()))
    us = 1e6 * dur / np.maximum(ne, 1)
    print('us/exp percentiles 5/50/95', np.percentile(us, [5, 50, 95]), 'weighted mean', 1e6 * dur.sum() / ne.sum())
    idx = np.argsort(-dur)[:8]
    for i in idx:
        print('  long query', i, 'exp', int(ne[i]), 'dur', round(dur[i], 3), 'begin', round(T[i, 0], 3), 'us/exp', round(us[i], 2), 'refill', R[i].n_refill, 'evict', R[i].n_evict,
              {k: round(v / ne[i]) for k, v in pl.queryCycles(int(i)).items()})
    for i in np.where(np.array([r.status for r in R]) == 4)[0]:
        print('  POOL_FULL query', i, 'exp', R[i].n_expanded, 'nodes', R[i].n_nodes, 'edges', R[i].n_edges, 'push', R[i].n_push)
    print('  totals nodes', sum(r.n_nodes for r in R), 'edges', sum(r.n_edges for r in R), 'push', sum(r.n_push for r in R))
    # concurrency over time
    for frac in (0.1, 0.25, 0.5, 0.75, 0.9):
        tt = frac * T[:, 1].max()
        print('  t=%.2f running %d' % (tt, int(((T[:, 0] <= tt) & (T[:, 1] > tt)).sum())))
if which == 'width':
    grid, origin, res, start, goal, rng = mapgen.benchmark_map(256)
    U = mapgen.control_lattice(1.0, 1, True)
    for wmul in (2.0, 4.0, 8.0, 16.0, 32.0):
        mu, pl = util.make_gpu(grid, origin, res, U, v_max=2.0, a_max=1.0, max_nodes=1 << 22, max_edges=1 << 24, max_log=1 << 23, spec=SPEC)
        pl.setBucketWidth(10.0 * wmul)
        ok = pl.plan(util.gpu_wp(start), util.gpu_wp(goal)); r = pl.getResult()
        cy = pl.queryCycles()
        print('width', 10.0 * wmul, 'us/exp', round(1e3 * pl.lastKernelMs() / r.n_expanded, 3), 'refill', r.n_refill, 'evict', r.n_evict, {k: round(v / r.n_expanded) for k, v in cy.items() if k in ('pop', 'expand', 'lookup', 'commit', 'refill', 'evict')}, 'exp/batch', round(r.n_expanded / max(cy['batches'], 1), 2), r.n_expanded)
