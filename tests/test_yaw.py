"""Yaw-carrying search states (use_yaw lattices, map_planner_node.cpp:119-139,165,179): the CPU oracle's restatement and,
with -m gpu, the HIP search through the C-ABI against it -- bit-exact, the whole state space.

The reference has no test, launch file or golden vector that turns use_yaw on (both test.launch files pass
use_yaw = false), and the yaw arithmetic lives in the absent motion_primitive_library submodule: the yaw rules are the
[UNVERIFIED] restatement listed in oracle/mpl_oracle.h, so the pins below are self-consistency checks and regression
values of that restatement, not reference outputs."""
import json
import os

import numpy as np
import pytest

from mpl_ros_amd import mapgen
from oracle import orc
from tests import util

HERE = os.path.dirname(os.path.abspath(__file__))
SIMPLE_START, SIMPLE_GOAL = (14.5, 4.5, 0.05), (2.4, 16.6, 0.05)


@pytest.fixture(scope="module")
def simple_map():
    d = np.load(os.path.join(HERE, "golden", "simple_map.npz"))
    return d["grid"], tuple(d["origin"].tolist()), float(d["res"])


def test_yaw_lattice_is_the_reference_drivers():
    U = mapgen.control_lattice(1.0, 1, False, u_yaw=0.5)
    assert U.shape == (27, 4) and U[0].tolist() == [-1, -1, 0, -0.5] and U[1].tolist() == [-1, -1, 0, 0] and U[26].tolist() == [1, 1, 0, 0.5]
    assert mapgen.control_lattice(1.0, 1, True, u_yaw=0.5).shape == (81, 4)
    assert np.array_equal(U[::3, :3], mapgen.control_lattice(1.0, 1, False))


def test_oracle_yaw_search_without_constraint_is_the_yawless_search(simple_map):
    """No yaw input (rate 0 everywhere), no yaw threshold, equal start / goal yaw: the yaw key never changes, so the
    search must expand exactly what the yaw-less search expands."""
    grid, origin, res = simple_map
    U3 = mapgen.control_lattice(1.0, 1, False)
    U4 = np.concatenate([U3, np.zeros((len(U3), 1))], axis=1)
    kw = dict(v_max=2.0, a_max=1.0, tol_pos=0.5)
    P3 = util.make_oracle(grid, origin, res, orc.ACC, U3, **kw)
    P4 = util.make_oracle(grid, origin, res, orc.ACC | orc.YAW, U4, **kw)
    assert P3.plan(orc.waypoint(SIMPLE_START), orc.waypoint(SIMPLE_GOAL)) == 0
    assert P4.plan(orc.waypoint(SIMPLE_START, yaw=0.3), orc.waypoint(SIMPLE_GOAL, yaw=0.3)) == 0
    assert P4.traj_cost == P3.traj_cost and np.array_equal(P4.expanded()[0], P3.expanded()[0])
    assert all(w.yaw == 0.3 and w.control == (orc.ACC | orc.YAW) for w in P4.traj()["wps"])


def test_oracle_yaw_launch_query_regression_pin(simple_map):
    """The config-1 launch query with use_yaw = true (27 inputs, yaw_max = 0.5): every primitive of the result satisfies
    the yaw constraint at both ends, and the search is pinned (a regression value of the restatement)."""
    grid, origin, res = simple_map
    U = mapgen.control_lattice(1.0, 1, False, u_yaw=0.5)
    P = util.make_oracle(grid, origin, res, orc.ACC | orc.YAW, U, v_max=2.0, a_max=1.0, tol_pos=0.5, yaw_max=0.5)
    assert P.plan(orc.waypoint(SIMPLE_START, yaw=0.0), orc.waypoint(SIMPLE_GOAL, yaw=0.0)) == 0
    assert (P.traj_cost, len(P.expanded()[0]), P.num_nodes()) == (128.0, 1108, 2259)
    tr = P.traj()
    for w in tr["wps"]:
        v = np.array(w.vel[:2])
        if np.any(v != 0):
            assert v @ np.array([np.cos(w.yaw), np.sin(w.yaw)]) / np.linalg.norm(v) >= np.cos(0.5) - 1e-12
        assert -np.pi <= w.yaw <= np.pi
    # the constraint costs something: the yaw-less plan of the same query is cheaper (tests/golden/simple_plan.json)
    assert P.traj_cost > json.load(open(os.path.join(HERE, "golden", "simple_plan.json")))["cost"]


def test_yaw_api_without_a_device():
    """setU takes the Vec4f lattices, setYawmax stores; a yaw-rate lattice over yaw-less start states is refused before
    anything touches the device (it would otherwise be planned as a different search)."""
    from mpl_ros_amd._capi import MplxError
    from mpl_ros_amd.planner import VoxelMapPlanner, Waypoint3D
    pl = VoxelMapPlanner(False)
    pl.setU(mapgen.control_lattice(1.0, 1, False, u_yaw=0.5))
    pl.setYawmax(0.5)
    s = Waypoint3D(orc.ACC)
    assert not s.use_yaw
    with pytest.raises(MplxError):
        pl.plan(s, Waypoint3D(orc.ACC))
    s.use_yaw = True
    assert s.control == (orc.ACC | 16) and Waypoint3D(s.control).use_yaw  # `Waypoint3D goal(start.control)`, map_planner_node.cpp:167


# ------------------------------------------------------------------ HIP vs oracle
@pytest.mark.gpu
@pytest.mark.parametrize("control,use_3d", [(orc.ACC, True), (orc.ACC, False), (orc.JRK, False), (orc.VEL, True), (orc.SNP, False)])
def test_expand_batch_with_yaw_matches_get_succ(control, use_3d):
    grid, origin, res = util.small_map(64)
    U = mapgen.control_lattice(1.0, 1, use_3d, u_yaw=0.5)
    kw = dict(v_max=2.0, a_max=1.0, yaw_max=0.5)
    if control in (orc.JRK, orc.SNP):
        kw["j_max"] = 1.0
    P = util.make_oracle(grid, origin, res, control | orc.YAW, U, **kw)
    mu, pl = util.make_gpu(grid, origin, res, U, **kw)
    rng = np.random.default_rng(77 + control)
    states = util.random_states(rng, 300, control, 0.3, 6.1)
    yaws = rng.uniform(-np.pi, np.pi, len(states))
    # half of the states head along their planar velocity (the constraint then passes now and again)
    for i, (p, v, a, j) in enumerate(states):
        if i % 2 == 0 and (v[0] != 0 or v[1] != 0):
            yaws[i] = np.arctan2(v[1], v[0]) + rng.uniform(-0.6, 0.6)
    nodes = [util.gpu_wp(p, v, a, j, control, t=0.5 * i, yaw=yaws[i]) for i, (p, v, a, j) in enumerate(states)]
    out = pl.getSuccBatch(nodes)
    nU = U.shape[0]
    n_valid = n_free = 0
    for k, (p, v, a, j) in enumerate(states):
        cur = orc.waypoint(p, v, a, j, control, t=0.5 * k, yaw=yaws[k])
        succ, cost, act = P.get_succ(cur)
        got_valid = [out[k * nU + i] for i in range(nU) if out[k * nU + i].valid]
        assert [g.action for g in got_valid] == list(act)
        n_valid += len(act)
        for g, so, co in zip(got_valid, succ, cost):
            assert g.cost == co or (np.isinf(g.cost) and np.isinf(co))
            n_free += int(np.isfinite(co))
            for f in ("pos", "vel", "acc", "jrk"):
                assert np.array_equal(np.array(getattr(g.wp, f)[:]), np.array(getattr(so, f)[:]))
            assert g.wp.yaw == so.yaw and g.wp.t == so.t and g.wp.control == so.control == (control | orc.YAW)
            key = (orc.C.c_int32 * 16)()
            nk = orc.lib().orc_waypoint_key(orc.C.byref(so), key)
            m = min(nk, 12)  # (mplx_succ.key holds 12 integers: the yaw key of an SNP state is its 13th, reported through wp.yaw)
            assert g.nkey == nk and list(g.key[:m]) == list(key[:m])
    assert 0 < n_free <= n_valid < len(states) * nU  # the yaw constraint rejected some, passed some


@pytest.mark.gpu
def test_hip_plans_the_launch_query_with_use_yaw(simple_map):
    grid, origin, res = simple_map
    U = mapgen.control_lattice(1.0, 1, False, u_yaw=0.5)
    kw = dict(v_max=2.0, a_max=1.0, tol_pos=0.5, yaw_max=0.5)
    P = util.make_oracle(grid, origin, res, orc.ACC | orc.YAW, U, **kw)
    mu, pl = util.make_gpu(grid, origin, res, U, **kw)
    r, c = util.compare_plan(P, pl, (SIMPLE_START, (0, 0, 0)), (SIMPLE_GOAL,), orc.ACC, yaw=(0.0, 0.0))
    assert (r.status, r.cost, r.n_expanded) == (0, 128.0, 1108)
    assert pl.kernelName() == "astar_spec_kernel<32,16,ACC,yaw>"  # round 4: the YAW build of the speculative kernel (27-input lattice)
    ms_spec = pl.lastKernelMs()
    pl.setSpeculation(0)  # ... and the one-node kernel: the same search, whole state space
    r1, _ = util.compare_plan(P, pl, (SIMPLE_START, (0, 0, 0)), (SIMPLE_GOAL,), orc.ACC, yaw=(0.0, 0.0))
    assert pl.kernelName() == "astar_kernel<64,ACC,yaw>" and (r1.cost, r1.n_expanded, r1.expand_hash, r1.n_nodes) == (r.cost, r.n_expanded, r.expand_hash, r.n_nodes)
    print(f"yaw plan, {r.n_expanded} expansions: speculative kernel {ms_spec:.3f} ms, one-node kernel {pl.lastKernelMs():.3f} ms")
    pl.setSpeculation(-1)
    r, c = util.compare_plan(P, pl, (SIMPLE_START, (0, 0, 0)), (SIMPLE_GOAL,), orc.ACC, yaw=(0.0, 0.0))
    # the state space's states carry their yaw
    coords = pl._nodes()[0]
    for i in (0, 1, r.n_nodes // 2, r.n_nodes - 1):
        assert coords[i].yaw == P.node(i)[0].yaw and coords[i].control == (orc.ACC | orc.YAW)
    # blocked primitives / hm_.size() re-derived with the yaw key taking part
    po, ao = P.blocked_edges()
    pg, ag, n_all = pl.getBlockedEdges()
    assert sorted(zip(pg.tolist(), ag.tolist())) == sorted(zip(po.tolist(), ao.tolist())) and n_all == P.num_states_all()


@pytest.mark.gpu
@pytest.mark.parametrize("control,goal_yaw,yaw_max", [(orc.ACC, 1.0, 0.7), (orc.ACC, 0.0, -1.0), (orc.JRK, -2.0, 1.0), (orc.VEL, 0.5, 0.5), (orc.SNP, 0.3, 0.8)])
def test_hip_yaw_plans_in_3d(control, goal_yaw, yaw_max):
    """81-input (x, y, z, yaw rate) lattice, start / goal yaw different (the goal's yaw key only matters to the
    heuristic's `state == goal` shortcut; reaching it is decided by the position tolerance)."""
    grid, origin, res = util.small_map(64, seed=11, occupancy=0.06)
    mapgen.carve_bubble(grid, (1.05, 1.05, 1.05), origin, res, 3)
    mapgen.carve_bubble(grid, (4.55, 4.05, 3.05), origin, res, 3)
    U = mapgen.control_lattice(1.0, 1, True, u_yaw=0.5)
    kw = dict(v_max=2.0, a_max=1.0, yaw_max=yaw_max, max_expand=4000)
    if control in (orc.JRK, orc.SNP):  # (SNP + yaw: 13 key integers, 14 state doubles -- the record's last bytes)
        kw["j_max"] = 1.0
    P = util.make_oracle(grid, origin, res, control | orc.YAW, U, **kw)
    mu, pl = util.make_gpu(grid, origin, res, U, max_nodes=1 << 21, max_edges=1 << 23, **kw)
    start = ((1.05, 1.05, 1.05), (0, 0, 0), (0, 0, 0))
    r, c = util.compare_plan(P, pl, start, ((4.55, 4.05, 3.05),), control, yaw=(0.4, goal_yaw))
    assert r.n_expanded > 5


@pytest.mark.gpu
def test_yaw_and_yawless_plans_alternate_on_one_context(simple_map):
    """One MapUtil (= one device context), re-configured between a yaw-carrying and a yaw-less search: same record layout,
    nothing of the previous configuration leaks."""
    grid, origin, res = simple_map
    U3, U4 = mapgen.control_lattice(1.0, 1, False), mapgen.control_lattice(1.0, 1, False, u_yaw=0.5)
    kw = dict(v_max=2.0, a_max=1.0, tol_pos=0.5)
    P3 = util.make_oracle(grid, origin, res, orc.ACC, U3, **kw)
    P4 = util.make_oracle(grid, origin, res, orc.ACC | orc.YAW, U4, yaw_max=0.5, **kw)
    mu, pl = util.make_gpu(grid, origin, res, U3, yaw_max=0.5, **kw)
    for _ in range(2):
        pl.setU(U3)
        util.compare_plan(P3, pl, (SIMPLE_START, (0, 0, 0)), (SIMPLE_GOAL,), orc.ACC)
        pl.setU(U4)
        util.compare_plan(P4, pl, (SIMPLE_START, (0, 0, 0)), (SIMPLE_GOAL,), orc.ACC, yaw=(0.0, 0.0))


@pytest.mark.gpu
def test_lpastar_over_yaw_states_is_refused(simple_map):
    from mpl_ros_amd._capi import MplxError
    grid, origin, res = simple_map
    mu, pl = util.make_gpu(grid, origin, res, mapgen.control_lattice(1.0, 1, False, u_yaw=0.5), v_max=2.0, a_max=1.0, yaw_max=0.5)
    pl.setLPAstar(True)
    with pytest.raises(MplxError):
        pl.plan(util.gpu_wp(SIMPLE_START, yaw=0.0), util.gpu_wp(SIMPLE_GOAL, yaw=0.0))


def test_deterministic_sincos_is_accurate_and_validate_yaw_follows_its_definition():
    """det_sincos (the one + - * / sequence the oracle and the device both evaluate, so that `d < cos(yaw_max)` falls the
    same way on host and GPU) against libm over [-4 pi, 4 pi]; validate_yaw against its definition on hand-built
    primitives: heading along the motion passes, across it fails, a resting end is not tested, yaw_max <= 0 tests nothing."""
    L = orc.lib()
    sn, cs = orc.C.c_double(), orc.C.c_double()
    worst = 0.0
    for x in np.linspace(-4 * np.pi, 4 * np.pi, 20001):
        L.orc_det_sincos(float(x), orc.C.byref(sn), orc.C.byref(cs))
        worst = max(worst, abs(sn.value - np.sin(x)), abs(cs.value - np.cos(x)))
    assert worst < 3e-16

    def prim(vel, u, yaw, u_yaw):
        w = orc.waypoint((0, 0, 0), vel=vel, yaw=yaw)
        p = orc.Primitive()
        L.orc_primitive_build_yaw(orc.C.byref(w), (orc.C.c_double * 3)(*u), float(u_yaw), 1.0, orc.C.byref(p))
        return p

    assert L.orc_validate_yaw(orc.C.byref(prim((1, 0, 0), (0, 0, 0), 0.0, 0.0)), 0.5) == 1       # heading = direction of motion
    assert L.orc_validate_yaw(orc.C.byref(prim((1, 0, 0), (0, 0, 0), 0.4, 0.0)), 0.5) == 1       # 0.4 rad off: inside 0.5
    assert L.orc_validate_yaw(orc.C.byref(prim((1, 0, 0), (0, 0, 0), 0.6, 0.0)), 0.5) == 0       # 0.6 rad off: outside
    assert L.orc_validate_yaw(orc.C.byref(prim((1, 0, 0), (0, 0, 0), 0.4, 0.2)), 0.5) == 0       # drifts to 0.6 by the end
    assert L.orc_validate_yaw(orc.C.byref(prim((0, 0, 0), (0, 1, 0), 0.0, 0.0)), 0.5) == 0       # starts at rest (not tested), ends moving along +y with yaw 0
    assert L.orc_validate_yaw(orc.C.byref(prim((0, 0, 0), (0, 1, 0), 1.5, 0.0)), 0.5) == 1       # ... with yaw ~ pi / 2
    assert L.orc_validate_yaw(orc.C.byref(prim((0, 1, 0), (0, 0, 0), 0.0, 0.0)), -1.0) == 1      # no threshold
