"""-m gpu: HIP path (through the C-ABI) vs the CPU oracle on the same seeded inputs. Bit-exact."""
import numpy as np
import pytest

from mpl_ros_amd import mapgen
from oracle import orc
from tests import util

pytestmark = pytest.mark.gpu

CFG = {
    orc.ACC: dict(num=1, kw=dict(v_max=2.0, a_max=1.0)),
    orc.JRK: dict(num=2, kw=dict(v_max=2.0, a_max=1.0, j_max=1.0)),
}


@pytest.mark.parametrize("control", [orc.ACC, orc.JRK])
def test_expand_batch_matches_get_succ(control):
    grid, origin, res = util.small_map(64)
    U = mapgen.control_lattice(1.0, CFG[control]["num"], True)
    kw = CFG[control]["kw"]
    P = util.make_oracle(grid, origin, res, control, U, **kw)
    mu, pl = util.make_gpu(grid, origin, res, U, **kw)
    rng = np.random.default_rng(1234 + control)
    states = util.random_states(rng, 200, control, 0.3, 6.1)
    nodes = [util.gpu_wp(p, v, a, j, control, t=0.5 * i) for i, (p, v, a, j) in enumerate(states)]
    out = pl.getSuccBatch(nodes)
    nU = U.shape[0]
    total_reads = 0
    P.reset_counters()
    for k, (p, v, a, j) in enumerate(states):
        cur = orc.waypoint(p, v, a, j, control, t=0.5 * k)
        succ, cost, act = P.get_succ(cur)
        got = [out[k * nU + i] for i in range(nU)]
        got_valid = [g for g in got if g.valid]
        assert [g.action for g in got_valid] == list(act)
        for g, so, co in zip(got_valid, succ, cost):
            assert g.cost == co or (np.isinf(g.cost) and np.isinf(co))
            assert np.array_equal(np.array(g.wp.pos[:]), np.array(so.pos[:]))
            assert np.array_equal(np.array(g.wp.vel[:]), np.array(so.vel[:]))
            assert np.array_equal(np.array(g.wp.acc[:]), np.array(so.acc[:]))
            assert np.array_equal(np.array(g.wp.jrk[:]), np.array(so.jrk[:]))
            assert g.wp.t == so.t
            key = (orc.C.c_int32 * 13)()
            so.control = control
            nk = orc.lib().orc_waypoint_key(orc.C.byref(so), key)
            assert g.nkey == nk and list(g.key[:nk]) == list(key[:nk])
        total_reads += sum(g.voxel_reads for g in got_valid)
    assert total_reads == P.counters()["n_voxel_reads"]


@pytest.mark.parametrize("control", [orc.ACC, orc.JRK])
def test_heuristic_and_goal(control):
    grid, origin, res = util.small_map(64)
    U = mapgen.control_lattice(1.0, 1, True)
    kw = CFG[control]["kw"]
    P = util.make_oracle(grid, origin, res, control, U, **kw)
    mu, pl = util.make_gpu(grid, origin, res, U, **kw)
    rng = np.random.default_rng(99)
    states = util.random_states(rng, 500, control, 0.0, 6.4)
    goal_o = orc.waypoint((5.05, 4.05, 3.05), control=control)
    P.set_goal(goal_o)
    h, isg = pl.heuristicBatch([util.gpu_wp(p, v, a, j, control) for (p, v, a, j) in states], util.gpu_wp((5.05, 4.05, 3.05), control=control))
    for i, (p, v, a, j) in enumerate(states):
        so = orc.waypoint(p, v, a, j, control)
        assert h[i] == P.heuristic(so)
        assert bool(isg[i]) == P.is_goal(so)


def test_map_query_matches_float_to_int(skir):
    grid, origin, res = skir
    P = util.make_oracle(grid, origin, res, orc.ACC, mapgen.control_lattice())
    mu, pl = util.make_gpu(grid, origin, res, mapgen.control_lattice())
    rng = np.random.default_rng(5)
    pts = rng.uniform(-0.5, 10.5, (2000, 3))
    pts[:100] = np.round(pts[:100], 1)  # cell-boundary cases
    cells, st = mu.query(pts)
    for i in range(len(pts)):
        assert tuple(cells[i]) == P.float_to_int(pts[i])
        assert (st[i] == 0) == P.is_free_point(pts[i])
    assert np.array_equal(mu.getMap().reshape(grid.shape), grid)


SPEC = pytest.mark.parametrize("spec", [0, 2], ids=["sequential", "speculative"])


@SPEC
def test_plan_skir_reference_query(skir, spec):
    """The one runnable reference scenario: maps/skir/skir.bag + launch/map_planner_node/test.launch.skir."""
    grid, origin, res = skir
    U = mapgen.control_lattice(1.0, 1, True)
    kw = dict(v_max=2.0, a_max=1.0, tol_pos=0.5)
    P = util.make_oracle(grid, origin, res, orc.ACC, U, **kw)
    mu, pl = util.make_gpu(grid, origin, res, U, record=1 << 16, spec=spec, **kw)
    r, c = util.compare_plan(P, pl, ((5.5, 5.5, 0.5), (1, 0, 0)), ((1.5, 1.5, 5.5),), orc.ACC)
    ids_o, pos_o = P.expanded()
    assert np.array_equal(pl.getExpandedIds(), ids_o)
    assert np.array_equal(pl.getExpandedNodes(), pos_o)
    assert len(pl.getCloseSet()) == P.num_closed()
    assert r.status == 0 and r.cost == 59.0 and r.n_expanded == 333


@SPEC
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_plan_acc_synthetic(seed, spec):
    grid, origin, res = util.small_map(96, seed=seed, occupancy=0.10)
    mapgen.carve_bubble(grid, (1.05, 1.05, 1.05), origin, res, 3)
    mapgen.carve_bubble(grid, (8.55, 8.55, 8.55), origin, res, 3)
    U = mapgen.control_lattice(1.0, 1, True)
    kw = dict(v_max=2.0, a_max=1.0, tol_pos=0.5)
    P = util.make_oracle(grid, origin, res, orc.ACC, U, **kw)
    mu, pl = util.make_gpu(grid, origin, res, U, spec=spec, **kw)
    util.compare_plan(P, pl, ((1.05, 1.05, 1.05), (0, 0, 0)), ((8.55, 8.55, 8.55),), orc.ACC)


@SPEC
def test_plan_jrk_capped(spec):
    grid, origin, res = util.small_map(96, seed=4, occupancy=0.10)
    mapgen.carve_bubble(grid, (1.05, 1.05, 1.05), origin, res, 3)
    U = mapgen.control_lattice(1.0, 2, True)
    kw = dict(v_max=2.0, a_max=1.0, j_max=1.0, tol_pos=0.5, max_expand=3000)
    P = util.make_oracle(grid, origin, res, orc.JRK, U, **kw)
    mu, pl = util.make_gpu(grid, origin, res, U, spec=spec, **kw)
    r, c = util.compare_plan(P, pl, ((1.05, 1.05, 1.05), (0, 0, 0), (0, 0, 0)), ((8.55, 8.55, 8.55),), orc.JRK)


@SPEC
def test_plan_jrk_reaches_goal(spec):
    grid, origin, res = util.small_map(64, seed=5, occupancy=0.05)
    mapgen.carve_bubble(grid, (1.05, 1.05, 1.05), origin, res, 3)
    mapgen.carve_bubble(grid, (4.55, 4.55, 3.05), origin, res, 3)
    U = mapgen.control_lattice(1.0, 1, True)
    kw = dict(v_max=2.0, a_max=1.0, j_max=1.0, tol_pos=0.5, max_expand=50000)
    P = util.make_oracle(grid, origin, res, orc.JRK, U, **kw)
    mu, pl = util.make_gpu(grid, origin, res, U, spec=spec, **kw)
    util.compare_plan(P, pl, ((1.05, 1.05, 1.05), (0, 0, 0), (0, 0, 0)), ((4.55, 4.55, 3.05),), orc.JRK)


@SPEC
def test_start_occupied_and_unreachable(spec):
    grid, origin, res = util.small_map(32, seed=9, occupancy=0.05)
    grid[:, :, :] = np.where(grid > 0, grid, 0)
    occ = np.argwhere(grid > 0)[0]
    U = mapgen.control_lattice(1.0, 1, True)
    kw = dict(v_max=2.0, a_max=1.0)
    P = util.make_oracle(grid, origin, res, orc.ACC, U, **kw)
    mu, pl = util.make_gpu(grid, origin, res, U, spec=spec, **kw)
    p_occ = ((occ[2] + 0.5) * res, (occ[1] + 0.5) * res, (occ[0] + 0.5) * res)
    util.compare_plan(P, pl, (p_occ, (0, 0, 0)), ((1.0, 1.0, 1.0),), orc.ACC)
    # goal walled in: OPEN runs empty (velocity-bounded lattice in a closed box is finite)
    g2 = np.zeros((24, 24, 24), dtype=np.int8)
    g2[8:16, 8:16, 8] = 100; g2[8:16, 8:16, 15] = 100
    g2[8:16, 8, 8:16] = 100; g2[8:16, 15, 8:16] = 100
    g2[8, 8:16, 8:16] = 100; g2[15, 8:16, 8:16] = 100
    P = util.make_oracle(g2, origin, res, orc.ACC, U, **kw)
    mu, pl = util.make_gpu(g2, origin, res, U, spec=spec, **kw)
    r, c = util.compare_plan(P, pl, ((1.15, 1.15, 1.15), (0, 0, 0)), ((0.25, 0.25, 0.25),), orc.ACC)
    assert r.status == 1


@SPEC
def test_duplicate_controls_take_the_ordered_path(spec):
    """Two identical control inputs give two successors with one key: the commit must serialise."""
    grid, origin, res = util.small_map(48, seed=11, occupancy=0.05)
    mapgen.carve_bubble(grid, (1.05, 1.05, 1.05), origin, res, 3)
    U = mapgen.control_lattice(1.0, 1, True)
    U = np.vstack([U, U[5:9], U[20:22] + 1e-4])  # exact and near duplicates
    kw = dict(v_max=2.0, a_max=1.0, max_expand=400)
    P = util.make_oracle(grid, origin, res, orc.ACC, U, **kw)
    mu, pl = util.make_gpu(grid, origin, res, U, spec=spec, **kw)
    util.compare_plan(P, pl, ((1.05, 1.05, 1.05), (0, 0, 0)), ((3.55, 3.55, 3.05),), orc.ACC)


@SPEC
@pytest.mark.parametrize("copies,jitter", [(3, 0.0), (2, 2e-4), (4, 1e-4)])
def test_many_lanes_on_one_state(spec, copies, jitter):
    """Every control input several times (exact copies, or copies that quantise to the same key with a
    slightly different cost): up to `copies` lanes of each unit, and lanes of several units, meet on one
    state.  The predecessor lists (compare_plan checks every node's list) must keep the sequential order."""
    grid, origin, res = util.small_map(48, seed=13, occupancy=0.06)
    mapgen.carve_bubble(grid, (1.05, 1.05, 1.05), origin, res, 3)
    U0 = mapgen.control_lattice(1.0, 1, True)
    U = np.vstack([U0 + k * jitter for k in range(copies)])
    kw = dict(v_max=2.0, a_max=1.0, max_expand=600)
    P = util.make_oracle(grid, origin, res, orc.ACC, U, **kw)
    mu, pl = util.make_gpu(grid, origin, res, U, spec=spec, **kw)
    util.compare_plan(P, pl, ((1.05, 1.05, 1.05), (0, 0, 0)), ((3.55, 3.05, 3.55),), orc.ACC)


@SPEC
def test_plan_batch_matches_single_queries(spec):
    grid, origin, res = util.small_map(96, seed=21, occupancy=0.10)
    U = mapgen.control_lattice(1.0, 1, True)
    kw = dict(v_max=2.0, a_max=1.0)
    P = util.make_oracle(grid, origin, res, orc.ACC, U, **kw)
    rng = mapgen.SplitMix64(77)
    queries = mapgen.random_queries(grid, origin, res, 24, rng, min_dist=4.0)
    mu, pl = util.make_gpu(grid, origin, res, U, n_slots=8, record=1 << 15, spec=spec, **kw)
    starts = [util.gpu_wp(s) for s, g in queries]
    goals = [util.gpu_wp(g) for s, g in queries]
    res_b = pl.planBatch(starts, goals)
    for q, (s, g) in enumerate(queries):
        st = P.plan(orc.waypoint(s), orc.waypoint(g))
        ids_o, _ = P.expanded()
        r = res_b[q]
        assert r.status == st
        assert r.n_expanded == len(ids_o) and r.expand_hash == util.expand_hash(ids_o)
        assert np.array_equal(pl.getExpandedIds(q), ids_o[: 1 << 15])
        if st == 0:
            assert r.cost == P.traj_cost
            to, tg = P.traj(), pl.getTraj(q)
            assert np.array_equal(tg.actions, to["actions"]) and np.array_equal(tg.node_ids, to["node_ids"])


@SPEC
def test_near_far_open_structure_under_pressure(spec):
    """Tiny bucket width / eps=0 (Dijkstra: massive f ties) stress refill, eviction and tie-breaking."""
    grid, origin, res = util.small_map(64, seed=31, occupancy=0.08)
    mapgen.carve_bubble(grid, (1.05, 1.05, 1.05), origin, res, 3)
    U = mapgen.control_lattice(1.0, 1, True)
    for eps, width, me in ((0.0, 0.0, 4000), (1.0, 1e-3, 6000), (1.0, 500.0, 6000), (3.0, 0.0, 6000)):
        kw = dict(v_max=2.0, a_max=1.0, eps=eps, max_expand=me)
        P = util.make_oracle(grid, origin, res, orc.ACC, U, **kw)
        mu, pl = util.make_gpu(grid, origin, res, U, spec=spec, **kw)
        pl.setBucketWidth(width)
        util.compare_plan(P, pl, ((1.05, 1.05, 1.05), (0, 0, 0)), ((5.55, 5.55, 5.05),), orc.ACC)


def test_occ_map_planner_2d():
    """OccMapPlanner (2-D) = the 3-D path with z frozen; checked against the oracle on the same embedding."""
    from mpl_ros_amd.planner import OccMapPlanner, OccMapUtil, Waypoint2D
    rng = np.random.default_rng(8)
    g2 = np.zeros((120, 160), dtype=np.int8)  # (dy, dx)
    for _ in range(60):
        x, y, w, h = rng.integers(5, 150), rng.integers(5, 110), rng.integers(2, 10), rng.integers(2, 10)
        g2[y:y + h, x:x + w] = 100
    g2[5:15, 5:15] = 0
    g2[100:112, 140:152] = 0
    res, origin2 = 0.1, (0.0, 0.0)
    U = mapgen.control_lattice(1.0, 1, False)  # 9 planar inputs (map_planner_node.cpp:116-118)
    mu = OccMapUtil()
    mu.setMap(origin2, (160, 120), g2.ravel(), res)
    pl = OccMapPlanner(False)
    pl.setMapUtil(mu)
    pl.setVmax(2.0); pl.setAmax(1.0); pl.setDt(1.0); pl.setTol(0.5)
    pl.setU(U[:, :2])
    pl.setCapacity(1, 1 << 20, 1 << 22, 1 << 21)
    s, g = Waypoint2D(orc.ACC), Waypoint2D(orc.ACC)
    s.pos[:2] = (1.05, 1.05)
    g.pos[:2] = (14.55, 10.55)
    ok = pl.plan(s, g)
    r = pl.getResult()
    P = util.make_oracle(g2.reshape(1, 120, 160), (0.0, 0.0, -0.05), res, orc.ACC, U, v_max=2.0, a_max=1.0, tol_pos=0.5)
    st = P.plan(orc.waypoint((1.05, 1.05, 0.0)), orc.waypoint((14.55, 10.55, 0.0)))
    ids_o, _ = P.expanded()
    assert ok == (st == 0) and r.status == st
    assert r.n_expanded == len(ids_o) and r.expand_hash == util.expand_hash(ids_o)
    assert r.cost == P.traj_cost
    tr = pl.getTraj()
    assert np.array_equal(tr.actions, P.traj()["actions"])
    assert all(w.pos[2] == 0.0 and w.vel[2] == 0.0 for w in tr.getWaypoints())
    cells, state = mu.query([(1.05, 1.05), (0.0, 0.0), (16.5, 3.0)])
    assert tuple(cells[0]) == (10, 10) and state[0] == 0 and state[1] == 3 and state[2] == 3


@SPEC
@pytest.mark.parametrize("seed", range(16))
def test_random_configurations(spec, seed):
    """Seeded sweep over what the setters expose: control kind, lattice size, dt, w, eps, limits, tolerances,
    t_max, heuristic kind, start velocity -- every plan compared with the oracle down to the state space."""
    rng = np.random.default_rng(1000 + seed)
    control = [orc.VEL, orc.ACC, orc.ACC, orc.JRK][seed % 4]
    num = int(rng.integers(1, 3)) if control != orc.JRK else 1
    use_3d = bool(rng.integers(0, 4) > 0)
    u = float(rng.choice([0.5, 1.0, 2.0]))
    U = mapgen.control_lattice(u, num, use_3d)
    n = 48
    grid, origin, res = util.small_map(n, seed=50 + seed, occupancy=float(rng.choice([0.03, 0.08, 0.15])))
    kw = dict(dt=float(rng.choice([0.5, 1.0])), w=float(rng.choice([1.0, 10.0, 30.0])), eps=float(rng.choice([0.0, 0.5, 1.0, 2.0])),
              v_max=float(rng.choice([1.0, 2.0, 3.0])), a_max=float(rng.choice([-1.0, 1.0, 2.0])), tol_pos=float(rng.choice([0.3, 0.5, 1.0])),
              max_expand=1500, heur_ignore_dynamics=bool(rng.integers(0, 5) == 0))
    if control == orc.JRK:
        kw["j_max"] = float(rng.choice([1.0, 2.0]))
    if rng.integers(0, 3) == 0:
        kw["tol_vel"] = 1.0
    if rng.integers(0, 4) == 0:
        kw["t_max"] = 6.0 * kw["dt"]
    z = 1.05 if use_3d else 2.45
    start_p, goal_p = (1.05, 1.05, z), (float(rng.choice([3.05, 3.55])), float(rng.choice([2.55, 3.55])), z if not use_3d else 2.55)
    mapgen.carve_bubble(grid, start_p, origin, res, 3)
    mapgen.carve_bubble(grid, goal_p, origin, res, 3)
    v0 = (float(rng.integers(-1, 2)), 0.0, 0.0) if control != orc.VEL else (0.0, 0.0, 0.0)
    P = util.make_oracle(grid, origin, res, control, U, **kw)
    mu, pl = util.make_gpu(grid, origin, res, U, spec=spec, **kw)
    start = (start_p, v0, (0.0, 0.0, 0.0)) if control == orc.JRK else (start_p, v0)
    r, c = util.compare_plan(P, pl, start, (goal_p,), control)
    print(f"seed {seed}: control {control} nU {len(U)} status {r.status} expanded {r.n_expanded} nodes {r.n_nodes} edges {r.n_edges} reopen {r.n_reopen}")


@SPEC
def test_pool_full_is_reported_and_the_context_stays_usable(spec):
    """Pools too small for the search: MPLX_PLAN_POOL_FULL (never a truncated result); afterwards the same
    context, re-sized, plans correctly -- the far-bucket heads were left clean by the aborted query."""
    grid, origin, res = util.small_map(96, seed=2, occupancy=0.10)
    mapgen.carve_bubble(grid, (1.05, 1.05, 1.05), origin, res, 3)
    mapgen.carve_bubble(grid, (8.55, 8.55, 8.55), origin, res, 3)
    U = mapgen.control_lattice(1.0, 1, True)
    kw = dict(v_max=2.0, a_max=1.0)
    P = util.make_oracle(grid, origin, res, orc.ACC, U, **kw)
    mu, pl = util.make_gpu(grid, origin, res, U, spec=spec, max_nodes=1 << 15, max_edges=1 << 16, max_log=1 << 15, **kw)
    s, g = ((1.05, 1.05, 1.05), (0, 0, 0)), ((8.55, 8.55, 8.55),)
    pl.setEpsilon(0.0)  # uninformed search: far more than one chunk (32768) of states
    ok = pl.plan(util.gpu_wp(s[0]), util.gpu_wp(g[0]))
    assert not ok and pl.getResult().status == 4 and np.isinf(pl.getResult().cost)
    pl.setEpsilon(1.0)
    for _ in range(2):  # same context: bigger pools, then the query twice
        pl.setCapacity(1, 1 << 20, 1 << 22, 1 << 21)
        util.compare_plan(P, pl, s, g, orc.ACC)


def test_one_context_many_configurations():
    """One MapUtil / planner pair reused across control kinds, lattices, maps and capacities (pools are
    re-created when the record size changes, bucket heads and tables are reused otherwise)."""
    from mpl_ros_amd.planner import VoxelMapPlanner, VoxelMapUtil
    mu, pl = VoxelMapUtil(), VoxelMapPlanner(False)
    pl.setMapUtil(mu)
    for it, (control, num, seed, cap) in enumerate([(orc.ACC, 1, 1, 1 << 20), (orc.JRK, 1, 2, 1 << 19), (orc.ACC, 2, 3, 1 << 21), (orc.VEL, 1, 4, 1 << 18),
                                                    (orc.ACC, 1, 5, 1 << 20), (orc.JRK, 2, 6, 1 << 20)]):
        grid, origin, res = util.small_map(64, seed=seed, occupancy=0.08)
        mapgen.carve_bubble(grid, (1.05, 1.05, 1.05), origin, res, 3)
        mapgen.carve_bubble(grid, (4.55, 4.05, 3.55), origin, res, 3)
        dz, dy, dx = grid.shape
        mu.setMap(origin, (dx, dy, dz), grid.ravel(), res)
        U = mapgen.control_lattice(1.0, num, True)
        kw = dict(v_max=2.0, a_max=1.0, max_expand=2000)
        if control == orc.JRK:
            kw["j_max"] = 1.0
        P = util.make_oracle(grid, origin, res, control, U, **kw)
        pl.setVmax(2.0); pl.setAmax(1.0); pl.setJmax(kw.get("j_max", -1.0)); pl.setDt(1.0); pl.setU(U); pl.setTol(0.5); pl.setMaxNum(2000)
        pl.setCapacity(1, cap, cap * 4, cap * 2)
        start = ((1.05, 1.05, 1.05), (0, 0, 0), (0, 0, 0)) if control == orc.JRK else ((1.05, 1.05, 1.05), (0, 0, 0))
        util.compare_plan(P, pl, start, ((4.55, 4.05, 3.55),), control)


@SPEC
@pytest.mark.parametrize("hid", [False, True])
def test_unlimited_velocity_heuristic(spec, hid):
    """v_max = -1 (the setters' default = unlimited): the heuristic drops the arrival-time bound instead of
    dividing by it; plans and the whole state space still match the oracle."""
    grid, origin, res = util.small_map(48, seed=17, occupancy=0.06)
    mapgen.carve_bubble(grid, (1.05, 1.05, 1.05), origin, res, 3)
    mapgen.carve_bubble(grid, (3.55, 3.05, 2.55), origin, res, 3)
    U = mapgen.control_lattice(1.0, 1, True)
    kw = dict(v_max=-1.0, a_max=-1.0, max_expand=3000, heur_ignore_dynamics=hid)
    P = util.make_oracle(grid, origin, res, orc.ACC, U, **kw)
    mu, pl = util.make_gpu(grid, origin, res, U, spec=spec, **kw)
    r, c = util.compare_plan(P, pl, ((1.05, 1.05, 1.05), (0, 0, 0)), ((3.55, 3.05, 2.55),), orc.ACC)
    assert r.status == 0
    h, _ = pl.heuristicBatch([util.gpu_wp((0.55, 0.55, 0.55)), util.gpu_wp((3.05, 3.05, 2.55))], util.gpu_wp((3.55, 3.05, 2.55)))
    assert h[0] > h[1] > 0


def test_two_planners_on_one_map_util_do_not_read_each_others_results():
    """The reference shares one MapUtil between planner_ and replan_planner_ (map_replanner_node.cpp:415,427).
    The device context keeps the LAST plan's state space: a planner whose results were overwritten refuses to
    answer (instead of returning the other planner's state space), and re-sends its own set-up when it plans."""
    from mpl_ros_amd._capi import MplxError
    from mpl_ros_amd.planner import VoxelMapPlanner
    grid, origin, res = util.small_map(48, seed=19, occupancy=0.06)
    mapgen.carve_bubble(grid, (1.05, 1.05, 1.05), origin, res, 3)
    U = mapgen.control_lattice(1.0, 1, True)
    mu, a = util.make_gpu(grid, origin, res, U, v_max=2.0, a_max=1.0, max_expand=500)
    b = VoxelMapPlanner(False)
    b.setMapUtil(mu)
    b.setVmax(1.0); b.setAmax(1.0); b.setDt(0.5); b.setU(U[:9]); b.setTol(0.5); b.setMaxNum(200)
    b.setCapacity(1, 1 << 20, 1 << 22, 1 << 21)
    s, g = util.gpu_wp((1.05, 1.05, 1.05)), util.gpu_wp((3.55, 3.05, 2.55))
    a.plan(s, g)
    ra = a.getResult().n_expanded
    ta = a.getTraj()
    b.plan(s, g)
    with pytest.raises(MplxError):
        a.getTraj()
    with pytest.raises(MplxError):
        a.getCloseSet()
    assert len(b.getCloseSet()) == b.getResult().n_closed
    a.plan(s, g)  # a's own set-up is sent again (b configured the shared context in between)
    assert a.getResult().n_expanded == ra
    tb = a.getTraj()
    assert np.array_equal(ta.actions, tb.actions) and tb.segs[0].t() == 1.0 if tb.segs else True


@pytest.mark.parametrize("control", [orc.ACC, orc.JRK])
def test_blocked_primitives_and_hm_size_match_upstream_accounting(control):
    """Upstream stores an hm_ entry and an inf-cost pred entry for every blocked successor (env_poly_map.h:60-66
    emits it).  The device keeps only finite arrivals and re-derives the rest on request: the set of (parent,
    action) pairs, the hm_.size() they imply and getAllPrimitives must equal the oracle's accounting (D7)."""
    grid, origin, res = util.small_map(48, seed=3, occupancy=0.12)
    mapgen.carve_bubble(grid, (1.05, 1.05, 1.05), origin, res, 3)
    U = mapgen.control_lattice(1.0, 1, True)
    kw = dict(v_max=2.0, a_max=1.0, max_expand=700)
    if control == orc.JRK:
        kw["j_max"] = 1.0
    P = util.make_oracle(grid, origin, res, control, U, **kw)
    mu, pl = util.make_gpu(grid, origin, res, U, **kw)
    start = ((1.05, 1.05, 1.05), (0, 0, 0), (0, 0, 0)) if control == orc.JRK else ((1.05, 1.05, 1.05), (0, 0, 0))
    r, c = util.compare_plan(P, pl, start, ((3.55, 3.55, 3.05),), control)
    po, ao = P.blocked_edges()
    pg, ag, n_all = pl.getBlockedEdges()
    assert len(pg) == c["n_succ"] - c["n_succ_finite"] == len(po) > 0
    assert sorted(zip(pg.tolist(), ag.tolist())) == sorted(zip(po.tolist(), ao.tolist()))
    assert n_all == P.num_states_all() > r.n_nodes
    assert len(pl.getAllPrimitives()) == r.n_edges + len(pg) and len(pl.getValidPrimitives()) == r.n_edges


def test_trajectory_longer_than_the_device_buffer_is_a_failed_plan_not_an_empty_success():
    """MPLX_PLAN_TRAJ_TOO_LONG: the goal is reached and the cost is right, but the path has more than 1024 primitives
    (the device-side recoverTraj buffer).  plan() must return False -- never True with an empty trajectory, which a
    replanner would execute -- while the search itself still equals the oracle's."""
    from mpl_ros_amd import _capi
    n = 1200
    grid = np.zeros((3, 3, n), dtype=np.int8)  # a corridor along x
    origin, res = (0.0, 0.0, 0.0), 0.1
    U = np.array([[1.0, 0.0, 0.0], [-1.0, 0.0, 0.0]])
    kw = dict(dt=0.1, tol_pos=0.05, w=10.0)
    P = util.make_oracle(grid, origin, res, orc.VEL, U, **kw)
    mu, pl = util.make_gpu(grid, origin, res, U, **kw)
    start, goal = (0.05, 0.15, 0.15), (0.05 + 110.0, 0.15, 0.15)
    st = P.plan(orc.waypoint(start, control=orc.VEL), orc.waypoint(goal, control=orc.VEL))
    assert st == orc.OK and P.traj()["n"] > 1024
    ok = pl.plan(util.gpu_wp(start, control=orc.VEL), util.gpu_wp(goal, control=orc.VEL))
    r = pl.getResult()
    assert r.status == _capi.PLAN_TRAJ_TOO_LONG and r.traj_len == 0
    assert r.cost == P.traj_cost and r.n_expanded == len(P.expanded()[0]) and r.expand_hash == util.expand_hash(P.expanded()[0])
    assert ok is False and np.isinf(pl.getTrajCost())
    assert len(pl.getTraj().segs) == 0
