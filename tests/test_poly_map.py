"""Moving-obstacle (PolyMap) environment, SURVEY.md 8 f1 / BASELINE config 5.

The checker here is the REFERENCE ITSELF: oracle/_ref/libpolymap_ref.so is the reference's env_poly_map.h +
poly_map_util.h + primitive_geometry_utils.h + simple_obstacle.h compiled from where they lie (the un-vendored
basis classes and DecompUtil's polyhedron come from include/mpl_shim).  -m gpu: the HIP kernels
(mplx_poly_get_succ_batch) against it, bit-exact, on random worlds and on the Team2 layout of robot_team.hpp."""
import numpy as np
import pytest

from mpl_ros_amd import poly_map as pm
from oracle import refpoly

needs_ref = pytest.mark.skipif(not refpoly.available(), reason="oracle/_ref/libpolymap_ref.so not built (make -C oracle ref)")

U9, acc_segs = pm.U9, pm.acc_segs


def random_world(rng, n_static=2, n_linear=2, n_nonlinear=3, dt=0.5):
    W = pm.PolyWorld((0.0, -5.0), (10.0, 10.0), start_t=float(rng.choice([0.0, 0.5, 1.25])))
    for _ in range(n_static):
        W.static.append(pm.StaticObstacle(pm.rectangle(float(rng.uniform(0.3, 1.0)), float(rng.uniform(0.3, 1.0))), rng.uniform((1, -4), (9, 4))))
    for _ in range(n_linear):
        W.linear.append(pm.LinearObstacle(pm.rectangle(0.5), rng.uniform((1, -4), (9, 4)), rng.uniform(-1, 1, 2), cov_v=float(rng.choice([0.0, 0.1]))))
    for _ in range(n_nonlinear):
        n = int(rng.integers(1, 7))
        us = U9[rng.integers(0, 9, n)]
        segs = acc_segs(rng.uniform((1, -4), (9, 4)), np.round(rng.uniform(-1, 1, 2), 1), us, dt)
        W.nonlinear.append(pm.NonlinearObstacle(pm.rectangle(0.5), segs, start_t=float(rng.choice([0.0, 0.3, -0.5, 2.0])),
                                                disappear_front=bool(rng.integers(0, 2)), disappear_back=bool(rng.integers(0, 2))))
    return W


def random_states(rng, n, dt=0.5):
    s = np.zeros((n, 9))
    s[:, 0:2] = np.round(rng.uniform((0.2, -4.8), (9.8, 4.8), (n, 2)), 2)
    s[:, 2:4] = np.round(rng.uniform(-2, 2, (n, 2)), 1)
    s[:, 8] = rng.integers(0, 8, n) * dt
    s[: n // 8, 0:2] = np.round(s[: n // 8, 0:2])  # lattice points: hyperplane-boundary cases
    return s


@needs_ref
def test_reference_environment_known_answers():
    """The compiled reference on hand-checkable cases (pins the stand-ins it was compiled against)."""
    W = pm.PolyWorld((0.0, -5.0), (10.0, 10.0))
    W.static.append(pm.StaticObstacle(pm.rectangle(1.0), (5.0, 0.0)))  # box [4,6] x [-1,1]
    R = refpoly.RefWorld(W, pm.ACC, U9, dt=1.0, v_max=2.0, a_max=1.0, w=10.0)
    succ, cost, act = R.get_succ([2.0, 0.0, 1.0, 0.0, 0, 0, 0, 0, 0.0])  # moving +x at 1 m/s towards the box
    by_act = dict(zip(act.tolist(), zip(succ, cost)))
    # u = (0, 0): ends at x = 3 (free, cost J_acc 0 + 0.001 * J_vel 1 + w dt 10); u = (1, 0): ends at 3.5, free
    assert by_act[4][0][0] == 3.0 and by_act[4][1] == 0.0 + 0.001 * 1.0 + 10.0
    assert np.isfinite(by_act[7][1])
    s2, c2, a2 = R.get_succ([3.5, 0.0, 1.0, 0.0, 0, 0, 0, 0, 0.0])  # 0.5 m before the box: every primitive enters it
    assert np.isinf(dict(zip(a2.tolist(), c2))[4]) and np.isinf(dict(zip(a2.tolist(), c2))[7])
    s3, c3, a3 = R.get_succ([0.2, 0.0, -1.0, 0.0, 0, 0, 0, 0, 0.0])  # leaves the bounding box for u_x <= 0
    assert 4 not in a3.tolist() and 1 not in a3.tolist()
    assert len(R.get_succ([9.0, 4.0, 2.0, 0.0, 0, 0, 0, 0, 0.0])[2]) < 9  # v_max: u_x = 1 would exceed 2 m/s


def _compare_get_succ(team, worlds, refs, world_of, states, n_u):
    out = team.get_succ_batch(world_of, states)
    n_inf = n_fin = 0
    for k, (w, s) in enumerate(zip(world_of, states)):
        succ, cost, act = refs[w].get_succ(s)
        got = [out[k * n_u + i] for i in range(n_u)]
        gv = [g for g in got if g.valid]
        assert [g.action for g in gv] == act.tolist(), (k, w)
        for g, so, co in zip(gv, succ, cost):
            assert np.array_equal(np.array(g.state[:]), so), (k, g.action)        # bit-exact f64
            assert g.cost == co or (np.isinf(g.cost) and np.isinf(co)), (k, g.action, g.cost, co)
            n_inf += int(np.isinf(co)); n_fin += int(np.isfinite(co))
    return n_fin, n_inf


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("seed", range(4))
def test_get_succ_matches_the_compiled_reference_on_random_worlds(seed):
    rng = np.random.default_rng(100 + seed)
    dt = 0.5
    worlds = [random_world(rng, dt=dt) for _ in range(6)]
    team = pm.PolyTeam()
    kw = dict(dt=dt, v_max=2.0, a_max=1.0, w=10.0)
    team.configure(pm.ACC, U9, **kw)
    team.set_worlds(worlds)
    refs = [refpoly.RefWorld(W, pm.ACC, U9, **kw) for W in worlds]
    K = 600
    world_of = rng.integers(0, len(worlds), K)
    states = random_states(rng, K, dt)
    n_fin, n_inf = _compare_get_succ(team, worlds, refs, world_of, states, 9)
    assert n_fin > 500 and n_inf > 300  # both outcomes are exercised


TEAM2 = pm.TEAM2


@pytest.mark.gpu
@needs_ref
def test_team2_tick_get_succ_matches_the_compiled_reference():
    """BASELINE config 5 at the expansion level: 16 robots of Team2, each seeing the 15 others as nonlinear obstacles
    (straight-line stand-in trajectories towards their goals) + the static box, dt 0.5, v_max 2, a_max 1."""
    dt = 0.5
    rng = np.random.default_rng(7)
    worlds, starts_t, _ = pm.team2_tick(dt=dt, t_now=1.0, traj_time=4.0)
    team = pm.PolyTeam()
    kw = dict(dt=dt, v_max=2.0, a_max=1.0, w=10.0)
    team.configure(pm.ACC, U9, **kw)
    team.set_worlds(worlds)
    refs = [refpoly.RefWorld(W, pm.ACC, U9, **kw) for W in worlds]
    K = 16 * 40
    world_of = np.repeat(np.arange(16), 40)
    states = random_states(rng, K, dt)
    for r in range(16):  # include each robot's own replanning start: traj.evaluate(dt)
        states[r * 40] = starts_t[r]
    n_fin, n_inf = _compare_get_succ(team, worlds, refs, world_of, states, 9)
    assert n_fin > 1000 and n_inf > 100


def _compare_plans(team, refs, world_of, starts, goals, compare_acc=False, **kw):
    cols = [0, 1, 2, 3, 4, 5, 8] if compare_acc else [0, 1, 2, 3, 8]  # (JRK states carry their acceleration)
    team.set_record(1 << 16)
    R = team.plan_batch(world_of, starts, goals, **kw)
    n_ok = 0
    for k, w in enumerate(world_of):
        ref = refs[w].plan(starts[k], goals[k], eps=kw.get("eps", 1.0), tol_pos=kw.get("tol_pos", 0.5), max_expand=kw.get("max_expand", -1),
                           heur_ignore_dynamics=kw.get("heur_ignore_dynamics", True))
        r = R[k]
        assert r.status == ref["status"], (k, r.status, ref["status"])
        assert r.n_expanded == len(ref["expanded"]) and r.n_nodes == ref["n_nodes"], (k, r.n_expanded, len(ref["expanded"]), r.n_nodes, ref["n_nodes"])
        assert np.array_equal(team.expanded_ids(k), ref["expanded"]), k  # same nodes in the same order
        if ref["status"] == 0:
            n_ok += 1
            assert r.cost == ref["cost"], (k, r.cost, ref["cost"])  # bit-exact f64
            act, ids, st = team.traj(k)
            assert np.array_equal(act, ref["actions"]) and np.array_equal(ids, ref["node_ids"]), k
            for i, nid in enumerate(ids):  # waypoint states: position, velocity and time of every node of the path
                s, _, _ = refs[w].node(int(nid))
                assert np.array_equal(st[i][cols], s[cols]), (k, i)
        else:
            assert np.isinf(r.cost)
    return R, n_ok


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("seed", range(3))
def test_plans_match_a_search_over_the_compiled_reference_environment(seed):
    """Whole plans: the device A* over the moving-obstacle environment against a best-first search (the restated
    GraphSearch loop) that expands through the REFERENCE's env_poly_map::get_succ -- expansion order, states created,
    cost, path actions / node ids / waypoint states, bit-exact."""
    rng = np.random.default_rng(300 + seed)
    dt = 0.5
    worlds = [random_world(rng, dt=dt) for _ in range(8)]
    team = pm.PolyTeam()
    kw = dict(dt=dt, v_max=2.0, a_max=1.0, w=10.0)
    team.configure(pm.ACC, U9, **kw)
    team.set_worlds(worlds)
    team.set_capacity(16, 1 << 20, 1 << 22, 1 << 21)
    refs = [refpoly.RefWorld(W, pm.ACC, U9, **kw) for W in worlds]
    n = 12
    world_of = rng.integers(0, len(worlds), n)
    starts, goals = np.zeros((n, 9)), np.zeros((n, 9))
    starts[:, 0:2] = np.round(rng.uniform((0.5, -4.5), (3.0, 4.5), (n, 2)), 1)
    starts[:, 8] = rng.integers(0, 3, n) * dt
    goals[:, 0:2] = np.round(rng.uniform((7.0, -4.5), (9.5, 4.5), (n, 2)), 1)
    R, n_ok = _compare_plans(team, refs, world_of, starts, goals, max_expand=4000)
    assert n_ok >= 4


@pytest.mark.gpu
@needs_ref
def test_team2_tick_plans_match():
    """BASELINE config 5: one decentralised tick of Team2 -- 16 robots replanning at once against each other's
    trajectories (4 s horizon) and the static box, one launch, against the search over the compiled reference env."""
    worlds, starts, goals = pm.team2_tick()
    team = pm.PolyTeam()
    kw = dict(dt=0.5, v_max=2.0, a_max=1.0, w=10.0)
    team.configure(pm.ACC, U9, **kw)
    team.set_worlds(worlds)
    team.set_capacity(16, 1 << 21, 1 << 23, 1 << 22)
    refs = [refpoly.RefWorld(W, pm.ACC, U9, **kw) for W in worlds]
    R, n_ok = _compare_plans(team, refs, np.arange(16), starts, goals, max_expand=20000)
    print("Team2 tick:", [(r.status, int(r.n_expanded)) for r in R], "kernel ms", team.last_kernel_ms())
    assert n_ok >= 12


@pytest.mark.gpu
@needs_ref
def test_team2_tick_plans_match_under_the_reference_parameters():
    """The same tick planned the way the reference plans it: robot.hpp:109-122 calls neither setHeurIgnoreDynamics nor
    setMaxNum, i.e. the dynamics-aware heuristic (env_base::cal_heur, ACC state -> ACC goal at rest) and no expansion cap.
    Every robot finds its trajectory (16 / 0); the device against the search over the compiled reference environment with
    the CPU oracle's heuristic, bit-exact as above."""
    worlds, starts, goals = pm.team2_tick()
    team = pm.PolyTeam()
    kw = dict(dt=0.5, v_max=2.0, a_max=1.0, w=10.0)
    team.configure(pm.ACC, U9, **kw)
    team.set_worlds(worlds)
    team.set_capacity(16, 1 << 21, 1 << 23, 1 << 22)
    refs = [refpoly.RefWorld(W, pm.ACC, U9, **kw) for W in worlds]
    R, n_ok = _compare_plans(team, refs, np.arange(16), starts, goals, max_expand=-1, heur_ignore_dynamics=False)
    print("Team2 tick, reference parameters:", [(r.status, int(r.n_expanded)) for r in R], "kernel ms", team.last_kernel_ms())
    assert n_ok == 16


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("seed", range(2))
def test_random_world_plans_match_with_the_dynamics_aware_heuristic(seed):
    """Random worlds, default (dynamics-aware) heuristic, goals at rest and moving goals, no cap unless a search runs away."""
    rng = np.random.default_rng(900 + seed)
    dt = 0.5
    worlds = [random_world(rng, dt=dt) for _ in range(8)]
    team = pm.PolyTeam()
    kw = dict(dt=dt, v_max=2.0, a_max=1.0, w=10.0)
    team.configure(pm.ACC, U9, **kw)
    team.set_worlds(worlds)
    team.set_capacity(16, 1 << 20, 1 << 22, 1 << 21)
    refs = [refpoly.RefWorld(W, pm.ACC, U9, **kw) for W in worlds]
    n = 12
    world_of = rng.integers(0, len(worlds), n)
    starts, goals = np.zeros((n, 9)), np.zeros((n, 9))
    starts[:, 0:2] = np.round(rng.uniform((0.5, -4.5), (3.0, 4.5), (n, 2)), 1)
    starts[:, 2:4] = np.round(rng.uniform(-1, 1, (n, 2)), 1)
    starts[:, 8] = rng.integers(0, 3, n) * dt
    goals[:, 0:2] = np.round(rng.uniform((7.0, -4.5), (9.5, 4.5), (n, 2)), 1)
    goals[n // 2:, 2:4] = np.round(rng.uniform(-1, 1, (n - n // 2, 2)), 1)  # (half of the goals carry a velocity)
    R, n_ok = _compare_plans(team, refs, world_of, starts, goals, max_expand=20000, heur_ignore_dynamics=False)
    assert n_ok >= 6


# ---------------------------------------------------------------- beyond VEL / ACC (round 3): the general solve()
# poly_map_planner_node.cpp:73-85 exposes use_acc / use_jrk (JRK / SNP robots); collide() then meets hyperplane equations
# of degree 3..5 and calls the general solve(a, b, c, d, e, f) (primitive_geometry_utils.h:28,71,148).  The device's
# statement of it is checked bit for bit on the host (tests/test_math_host.py); here the whole environment.
def random_world_general(rng, dt=0.5):
    """random_world plus obstacles that follow JRK trajectories (cubic segments: what a JRK robot's plan looks like)"""
    W = random_world(rng, dt=dt)
    for _ in range(3):
        n = int(rng.integers(1, 6))
        us = U9[rng.integers(0, 9, n)]
        segs = pm.jrk_segs(rng.uniform((1, -4), (9, 4)), np.round(rng.uniform(-1, 1, 2), 1), np.round(rng.uniform(-0.5, 0.5, 2), 1), us, dt)
        W.nonlinear.append(pm.NonlinearObstacle(pm.rectangle(0.5), segs, start_t=float(rng.choice([0.0, 0.3, -0.5, 2.0])),
                                                disappear_front=bool(rng.integers(0, 2)), disappear_back=bool(rng.integers(0, 2))))
    return W


def random_states_general(rng, n, control, dt=0.5):
    s = random_states(rng, n, dt)
    if control & 4:
        s[:, 4:6] = np.round(rng.uniform(-1, 1, (n, 2)), 1)
    if control & 8:
        s[:, 6:8] = np.round(rng.uniform(-1, 1, (n, 2)), 1)
    return s


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("control", [pm.JRK, pm.SNP, pm.ACC, pm.VEL])
def test_get_succ_of_any_control_kind_matches_the_compiled_reference(control):
    rng = np.random.default_rng(500 + control)
    dt = 0.5
    worlds = [random_world_general(rng, dt=dt) for _ in range(6)]
    team = pm.PolyTeam()
    kw = dict(dt=dt, v_max=2.0, a_max=1.0, j_max=1.5, w=10.0)
    team.configure(control, U9, **kw)
    team.set_worlds(worlds)
    refs = [refpoly.RefWorld(W, control, U9, **kw) for W in worlds]
    K = 500
    world_of = rng.integers(0, len(worlds), K)
    states = random_states_general(rng, K, control, dt)
    n_fin, n_inf = _compare_get_succ(team, worlds, refs, world_of, states, 9)
    assert n_fin > 200 and n_inf > 100  # both outcomes are exercised


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("control,general_obstacles", [(pm.JRK, True), (pm.JRK, False), (pm.ACC, True)])
def test_plans_beyond_acc_match_a_search_over_the_compiled_reference_environment(control, general_obstacles):
    """JRK states (keyed with position, velocity, acceleration and time) and / or obstacles on cubic trajectories: whole
    plans against the best-first search over the reference's env_poly_map::get_succ, bit-exact like the ACC case."""
    rng = np.random.default_rng(700 + control + int(general_obstacles))
    dt = 0.5
    worlds = [(random_world_general if general_obstacles else random_world)(rng, dt=dt) for _ in range(6)]
    team = pm.PolyTeam()
    kw = dict(dt=dt, v_max=2.0, a_max=1.0, w=10.0)
    team.configure(control, U9, **kw)
    team.set_worlds(worlds)
    team.set_capacity(16, 1 << 20, 1 << 22, 1 << 21)
    refs = [refpoly.RefWorld(W, control, U9, **kw) for W in worlds]
    n = 10
    world_of = rng.integers(0, len(worlds), n)
    starts, goals = np.zeros((n, 9)), np.zeros((n, 9))
    starts[:, 0:2] = np.round(rng.uniform((0.5, -4.5), (3.0, 4.5), (n, 2)), 1)
    starts[:, 8] = rng.integers(0, 3, n) * dt
    goals[:, 0:2] = np.round(rng.uniform((7.0, -4.5), (9.5, 4.5), (n, 2)), 1)
    R, n_ok = _compare_plans(team, refs, world_of, starts, goals, compare_acc=control == pm.JRK, max_expand=3000)
    assert n_ok >= 3


@pytest.mark.gpu
def test_snp_search_is_refused_loudly():
    team = pm.PolyTeam()
    team.configure(pm.SNP, U9, dt=0.5, v_max=2.0, a_max=1.0, j_max=1.0)
    W = pm.PolyWorld((0.0, -5.0), (10.0, 10.0))
    team.set_worlds([W])
    with pytest.raises(pm.MplxError):
        team.plan_batch([0], np.zeros((1, 9)), np.zeros((1, 9)))


@pytest.mark.gpu
@needs_ref
def test_lookahead_helpers_change_nothing_but_the_time():
    """The Team2 tick with the look-ahead helper workgroups off, with 3 and with the automatic number per robot: every
    result field, expansion order and path identical (the helpers only precompute a pure function of a state)."""
    worlds, starts, goals = pm.team2_tick()
    team = pm.PolyTeam()
    team.configure(pm.ACC, U9, dt=0.5, v_max=2.0, a_max=1.0, w=10.0)
    team.set_worlds(worlds)
    team.set_capacity(16, 1 << 21, 1 << 23, 1 << 22)
    team.set_record(1 << 16)
    runs = {}
    for h in (0, 3, -1):
        team.set_helpers(h)
        R = team.plan_batch(np.arange(16), starts, goals, max_expand=20000)
        assert team.last_helpers() == (h if h >= 0 else 4)
        runs[h] = [(r.status, r.n_expanded, r.n_nodes, r.n_edges, r.cost, r.expand_hash, r.traj_len) + tuple(team.traj(k)[0].tolist()) for k, r in enumerate(R)]
        for _ in range(2):  # repeatability of each configuration (leader rings, help masks: DESIGN's memory-ordering contract): three runs, equal
            R2 = team.plan_batch(np.arange(16), starts, goals, max_expand=20000)
            assert [(r.status, r.n_expanded, r.n_nodes, r.n_edges, r.cost, r.expand_hash, r.traj_len) for r in R2] == [t[:7] for t in runs[h]], h
        hits = [team.cycles(k)["lookahead_hits"] for k in range(16)]
        print("helpers", h, "kernel ms", team.last_kernel_ms(), "look-ahead hits of the longest robots", sorted(hits)[-4:])
        assert (sum(hits) > 0) == (h != 0)
    assert runs[0] == runs[3] == runs[-1]


@pytest.mark.gpu
@needs_ref
def test_more_queries_than_workgroups_run_without_helpers_and_match():
    """12 queries on 4 leader workgroups: every workgroup takes several queries one after the other, the look-ahead helpers
    stay off (they need one query per leader), the 256-lane kernel runs its own collision tests -- same results."""
    rng = np.random.default_rng(303)
    dt = 0.5
    worlds = [random_world(rng, dt=dt) for _ in range(5)]
    team = pm.PolyTeam()
    kw = dict(dt=dt, v_max=2.0, a_max=1.0, w=10.0)
    team.configure(pm.ACC, U9, **kw)
    team.set_worlds(worlds)
    team.set_capacity(4, 1 << 20, 1 << 22, 1 << 21)
    refs = [refpoly.RefWorld(W, pm.ACC, U9, **kw) for W in worlds]
    n = 12
    world_of = rng.integers(0, len(worlds), n)
    starts, goals = np.zeros((n, 9)), np.zeros((n, 9))
    starts[:, 0:2] = np.round(rng.uniform((0.5, -4.5), (3.0, 4.5), (n, 2)), 1)
    starts[:, 8] = rng.integers(0, 3, n) * dt
    goals[:, 0:2] = np.round(rng.uniform((7.0, -4.5), (9.5, 4.5), (n, 2)), 1)
    R, n_ok = _compare_plans(team, refs, world_of, starts, goals, max_expand=3000)
    assert team.last_helpers() == 0 and n_ok >= 3


# ---------------------------------------------------------------- the reference's own multi-robot loop, several ticks
def _device_planner(team, kw_plan):
    def plan_many(worlds, starts, goals):
        team.set_worlds(worlds)
        R = team.plan_batch(np.arange(len(worlds)), np.array(starts), np.array(goals), **kw_plan)
        out = []
        for k, r in enumerate(R):
            act, ids, st = team.traj(k)
            out.append((r.status, act.copy(), st.copy()))
        return out
    return plan_many


def _reference_planner(U, kw_env, kw_plan):
    def plan_many(worlds, starts, goals):
        out = []
        for W, s, g in zip(worlds, starts, goals):
            R = refpoly.RefWorld(W, pm.ACC, U, **kw_env)
            ref = R.plan(s, g, eps=kw_plan.get("eps", 1.0), tol_pos=kw_plan.get("tol_pos", 0.5), max_expand=kw_plan.get("max_expand", -1),
                         heur_ignore_dynamics=kw_plan.get("heur_ignore_dynamics", True))
            states = np.array([R.node(int(i))[0] for i in ref["node_ids"]]).reshape(-1, 9)
            states[:, 4:8] = 0.0  # (the device reports position, velocity and time of a path state)
            out.append((ref["status"], np.array(ref["actions"]), states))
        return out
    return plan_many


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("ddt,ref_params", [(0.01, False), (0.0, False), (0.0, True)])
def test_team2_decentralised_loop_over_many_ticks(ddt, ref_params):
    """multi_robot_node.cpp:95-105 with launch/multi_robot_node/test.launch (Team2, dt 0.5, v_max 2, a_max 1): Team2::init (16
    obstacle-free plans), then the 0.01 s loop for 1.1 s of simulated time -- every robot replans twice against the others'
    CURRENT planned trajectories.  ddt = 0.01 (the reference): one robot per tick; ddt = 0: all sixteen in the same tick (the
    batched tick of BASELINE config 5).  The device team and a team planning through the compiled reference environment
    must agree plan for plan (status, actions, path states) and therefore stay in the same state throughout."""
    kw_env = dict(dt=0.5, v_max=2.0, a_max=1.0, w=10.0)
    # ref_params: what robot.hpp:109-122 sets -- the dynamics-aware heuristic, no expansion cap; else round 3's capped
    # distance-heuristic variant
    kw_plan = dict(tol_pos=0.5, max_expand=-1, heur_ignore_dynamics=False) if ref_params else dict(tol_pos=0.5, max_expand=20000)
    dev = pm.PolyTeam()
    dev.configure(pm.ACC, U9, **kw_env)
    dev.set_capacity(16, 1 << 21, 1 << 23, 1 << 22)
    plan_dev, plan_ref = _device_planner(dev, kw_plan), _reference_planner(U9, kw_env, kw_plan)
    A, B = pm.RobotTeam(ddt=ddt), pm.RobotTeam(ddt=ddt)
    assert A.init(plan_dev, U9) and B.init(plan_ref, U9)
    n_replans = 0
    time = 0.0
    for tick in range(110):
        time += 0.01
        ok_a, due_a = A.update_decentralized(time, plan_dev, U9)
        ok_b, due_b = B.update_decentralized(time, plan_ref, U9)
        assert ok_a == ok_b and due_a == due_b
        n_replans += len(due_a)
        for ra, rb in zip(A.robots, B.robots):
            assert ra.traj_t == rb.traj_t and np.array_equal(ra.segs, rb.segs), tick
        if not ok_a:
            break
    assert n_replans >= 24 and A.plans == B.plans
