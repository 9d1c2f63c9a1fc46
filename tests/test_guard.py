"""-m gpu: the launch guard (include/mplx.h "Launch guard").  No entry point may wedge its caller -- the
reference's plan() always returns (mpl_test_node/src/map_planner_node.cpp:186-196).  A launch that does not end by itself
(test-only switch: workgroup 0 spins until it is told to stop) must come back as MPLX_ERR_TIMEOUT within the deadline plus
a little, name what the workgroups were doing, and leave the context usable: the next plan on it is bit-exact again."""
import time

import numpy as np
import pytest

from mpl_ros_amd import mapgen
from mpl_ros_amd._capi import MplxError
from oracle import orc
from tests import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kernel", ["speculative", "one-node"])
def test_a_launch_that_never_ends_comes_back_as_a_timeout(kernel):
    grid, origin, res = util.small_map(64, seed=11)
    U = mapgen.control_lattice(1.0, 1, True)
    kw = dict(v_max=2.0, a_max=1.0, tol_pos=0.5)
    mu, pl = util.make_gpu(grid, origin, res, U, spec=-1 if kernel == "speculative" else 0, **kw)
    start, goal = (0.55, 0.55, 0.55), (5.55, 5.55, 5.55)
    P = util.make_oracle(grid, origin, res, orc.ACC, U, **kw)
    r0, _ = util.compare_plan(P, pl, (start, (0, 0, 0)), (goal,), orc.ACC)  # (also: the context is configured and warm)
    pl.setDeadline(2.0)
    pl._debugHangNextLaunch()
    t0 = time.time()
    with pytest.raises(MplxError) as e:
        pl.plan(util.gpu_wp(start), util.gpu_wp(goal))
    dt = time.time() - t0
    assert 1.9 < dt < 12.0, dt
    assert "aborted" in str(e.value) and "test spin" in str(e.value), str(e.value)
    # the context survives: same query, same plan, bit for bit
    pl.setDeadline(60.0)
    r1, _ = util.compare_plan(P, pl, (start, (0, 0, 0)), (goal,), orc.ACC)
    assert (r1.status, r1.n_expanded, r1.expand_hash, r1.cost) == (r0.status, r0.n_expanded, r0.expand_hash, r0.cost)


def test_a_batch_that_outlives_its_deadline_is_aborted_and_the_next_one_is_clean():
    """A real search (no test switch): 64 long queries with a deadline far below their run time."""
    grid, origin, res, start, goal, _ = mapgen.benchmark_map(256)
    grid = np.ascontiguousarray(grid)
    U = mapgen.control_lattice(1.0, 1, True)
    kw = dict(v_max=2.0, a_max=1.0, tol_pos=0.5, max_expand=400_000)
    nq = 64
    queries = mapgen.c4_queries(grid, origin, res, nq, rank=1)
    pools = mapgen.c4_pools(False, nq, 400_000)
    mu, pl = util.make_gpu(grid, origin, res, U, n_slots=nq, max_nodes=pools["nodes"], max_edges=pools["edges"], max_log=pools["log"], **kw)
    S = [util.gpu_wp(s) for s, g in queries]
    G = [util.gpu_wp(g) for s, g in queries]
    word = lambda r: (r.status, r.traj_len, r.cost, r.n_expanded, r.n_nodes, r.n_edges, r.voxel_reads, r.expand_hash)
    ref = [word(r) for r in pl.planBatch(S, G)]
    ms = pl.lastKernelMs()
    if ms < 30.0:
        pytest.skip(f"the batch takes only {ms:.1f} ms: nothing to abort")
    pl.setDeadline(ms / 1000.0 / 4.0)
    t0 = time.time()
    with pytest.raises(MplxError) as e:
        pl.planBatch(S, G)
    assert time.time() - t0 < 10.0
    assert "aborted" in str(e.value), str(e.value)
    pl.setDeadline(60.0)
    assert [word(r) for r in pl.planBatch(S, G)] == ref


def test_the_moving_obstacle_tick_and_an_lpastar_plan_honour_their_deadlines():
    """The other two families of search launches behind the guard: the Team2 tick (16 leaders + look-ahead helper launch on a
    second stream, ~ 240 ms) and a fresh LPA* plan at 256^3 (speculative kernel + import, ~ 38 ms), each given a deadline far below its run time:
    MPLX_ERR_TIMEOUT, and the same call repeats exactly afterwards."""
    from mpl_ros_amd import poly_map as pm
    from mpl_ros_amd.planner import VoxelMapPlanner
    # ---- C5 tick under the reference's planner parameters
    U9 = pm.U9
    worlds, starts, goals = pm.team2_tick()
    team = pm.PolyTeam()
    team.configure(pm.ACC, U9, dt=0.5, v_max=2.0, a_max=1.0, w=10.0)
    team.set_worlds(worlds)
    team.set_capacity(16, 1 << 21, 1 << 23, 1 << 22)
    word = lambda r: (r.status, r.traj_len, r.cost, r.n_expanded, r.n_nodes, r.n_edges, r.expand_hash)
    ref = [word(r) for r in team.plan_batch(np.arange(16), starts, goals, max_expand=-1, heur_ignore_dynamics=False)]
    ms = team.last_kernel_ms()
    assert ms > 20.0, ms
    team.set_deadline(ms / 1000.0 / 5.0)
    t0 = time.time()
    with pytest.raises(MplxError) as e:
        team.plan_batch(np.arange(16), starts, goals, max_expand=-1, heur_ignore_dynamics=False)
    assert time.time() - t0 < 10.0 and "aborted" in str(e.value), str(e.value)
    team.set_deadline(60.0)
    assert [word(r) for r in team.plan_batch(np.arange(16), starts, goals, max_expand=-1, heur_ignore_dynamics=False)] == ref
    # ---- LPA*: a fresh plan of the C2 query
    grid, origin, res, start, goal, _ = mapgen.benchmark_map(256)
    U = mapgen.control_lattice(1.0, 1, True)
    mu, a = util.make_gpu(grid, origin, res, U, v_max=2.0, a_max=1.0, tol_pos=0.5)
    l = VoxelMapPlanner(False)
    l.setMapUtil(mu)
    l.setVmax(2.0); l.setAmax(1.0); l.setDt(1.0); l.setU(U); l.setTol(0.5)
    l.setCapacity(1, 1 << 20, 1 << 22, 1 << 22)
    l.setLPAstar(True)
    assert l.plan(util.gpu_wp(start), util.gpu_wp(goal))
    r0 = l.getResult()
    first = (r0.status, r0.cost, r0.n_expanded, r0.expand_hash)
    assert r0.n_expanded > 10000
    ms = l.lastKernelMs()  # (the plan's search launch: ~ 37 ms on the round-5 kernel)
    if ms < 8.0:
        pytest.skip(f"the LPA* plan's launch takes {ms:.1f} ms: too short to be given a quarter of its time")
    l.reset()
    l.setDeadline(ms / 1000.0 / 4.0)  # (derived from the measured launch, not a constant of one machine)
    t0 = time.time()
    with pytest.raises(MplxError) as e:
        l.plan(util.gpu_wp(start), util.gpu_wp(goal))
    assert time.time() - t0 < 10.0 and "aborted" in str(e.value), str(e.value)
    l.setDeadline(60.0)
    assert l.plan(util.gpu_wp(start), util.gpu_wp(goal))
    r1 = l.getResult()
    assert (r1.status, r1.cost, r1.n_expanded, r1.expand_hash) == first
