"""CPU: pins the oracle (oracle/mpl_oracle.c) with analytic known answers.

The reference holds no golden vectors for this path ("parity unpinned"), so the pins
are closed-form identities that follow from the in-tree polynomial convention
(primitive_geometry_utils.h:12-26) and an independent numerical optimal-control solution of the
heuristic's defining problem.
"""
import ctypes as C
from math import factorial

import numpy as np
import pytest

from mpl_ros_amd import mapgen
from oracle import orc

L = orc.lib()


def build(wp, u, dt):
    pr = orc.Primitive()
    L.orc_primitive_build(C.byref(wp), (C.c_double * 3)(*u), dt, C.byref(pr))
    return pr


def evaluate(pr, t):
    w = orc.Waypoint()
    L.orc_primitive_evaluate(C.byref(pr), t, C.byref(w))
    return w


def test_acc_primitive_from_rest():
    # ACC primitive from rest, u=(1,0,0), dt=1: pos +0.5, vel +1, J_acc = 1, J_vel = 1/3
    pr = build(orc.waypoint((0, 0, 0), control=orc.ACC), (1, 0, 0), 1.0)
    assert list(pr.c[0]) == [0, 0, 0, 1, 0, 0]
    w = evaluate(pr, 1.0)
    assert (w.pos[0], w.vel[0], w.acc[0], w.jrk[0]) == (0.5, 1.0, 1.0, 0.0)
    assert L.orc_primitive_J(C.byref(pr), orc.ACC) == 1.0
    assert abs(L.orc_primitive_J(C.byref(pr), orc.VEL) - 1.0 / 3) < 1e-15
    assert L.orc_primitive_max_vel(C.byref(pr), 0) == 1.0


def test_coefficient_layout_per_control():
    s = orc.waypoint((1, 2, 3), (4, 5, 6), (7, 8, 9), (10, 11, 12))
    for ctrl, expect in ((orc.VEL, [0, 0, 0, 0, .5, 1]), (orc.ACC, [0, 0, 0, .5, 4, 1]),
                         (orc.JRK, [0, 0, .5, 7, 4, 1]), (orc.SNP, [0, .5, 10, 7, 4, 1])):
        s.control = ctrl
        pr = build(s, (.5, .5, .5), 2.0)
        assert list(pr.c[0]) == expect


@pytest.mark.parametrize("ctrl", [orc.VEL, orc.ACC, orc.JRK, orc.SNP])
def test_evaluate_is_the_taylor_polynomial(ctrl):
    rng = np.random.default_rng(ctrl)
    for _ in range(20):
        p, v, a, j, u = (rng.uniform(-2, 2, 3) for _ in range(5))
        pr = build(orc.waypoint(p, v, a, j, control=ctrl), u, 1.0)
        t = rng.uniform(0, 1)
        w = evaluate(pr, t)
        for ax in range(3):
            c = np.array(pr.c[ax][:])
            mono = [c[5], c[4], c[3] / 2, c[2] / 6, c[1] / 24, c[0] / 120]
            for d, got in enumerate((w.pos[ax], w.vel[ax], w.acc[ax], w.jrk[ax])):
                ref = sum(mono[i] * factorial(i) / factorial(i - d) * t ** (i - d) for i in range(d, 6))
                assert abs(got - ref) < 1e-12


def test_max_abs_derivatives_bound_dense_sampling():
    # SNP is excluded on purpose: the restated extrema scan stops at the first root >= t while the
    # quadratic formula does not order its roots, so an interior extremum can be missed there
    # (recollected upstream behaviour, tagged UNVERIFIED in mpl_oracle.c).
    rng = np.random.default_rng(3)
    for ctrl in (orc.ACC, orc.JRK):
        for _ in range(50):
            p, v, a, j, u = (rng.uniform(-2, 2, 3) for _ in range(5))
            pr = build(orc.waypoint(p, v, a, j, control=ctrl), u, 1.0)
            ts = np.linspace(0, 1, 2001)
            for ax in range(3):
                vs = np.array([evaluate(pr, t).vel[ax] for t in ts[::20]])
                mv = L.orc_primitive_max_vel(C.byref(pr), ax)
                assert mv >= np.abs(vs).max() - 1e-12
                # JRK control: v is a parabola, the reported extremum is exact
                if ctrl == orc.JRK:
                    c = np.array(pr.c[ax][:])
                    cand = [abs(c[4]), abs(c[2] / 2 + c[3] + c[4])]
                    if c[2] != 0 and 0 < -c[3] / c[2] < 1:
                        tt = -c[3] / c[2]
                        cand.append(abs(c[2] / 2 * tt * tt + c[3] * tt + c[4]))
                    assert abs(mv - max(cand)) < 1e-12


def test_validate_primitive_rules():
    s = orc.waypoint((0, 0, 0), (1.9, 0, 0), control=orc.ACC)
    ok = build(s, (0.1, 0, 0), 1.0)
    bad = build(s, (0.2, 0, 0), 1.0)
    assert L.orc_validate_primitive(C.byref(ok), 2.0, 1.0, 1.0) == 1
    assert L.orc_validate_primitive(C.byref(bad), 2.0, 1.0, 1.0) == 0
    assert L.orc_validate_primitive(C.byref(bad), -1.0, 1.0, 1.0) == 1  # limit <= 0 disables the check
    sj = orc.waypoint((0, 0, 0), (0, 0, 0), (0.9, 0, 0), control=orc.JRK)
    assert L.orc_validate_primitive(C.byref(build(sj, (0.2, 0, 0), 1.0)), 2.0, 1.0, 1.0) == 0  # acc 1.1 > 1
    assert L.orc_validate_primitive(C.byref(build(sj, (0.1, 0, 0), 1.0)), 2.0, 1.0, 1.0) == 1


def test_J_matches_quadrature():
    rng = np.random.default_rng(11)
    for ctrl, d in ((orc.VEL, 1), (orc.ACC, 2), (orc.JRK, 3), (orc.SNP, 4)):
        p, v, a, j, u = (rng.uniform(-2, 2, 3) for _ in range(5))
        pr = build(orc.waypoint(p, v, a, j, control=orc.SNP), u, 0.7)
        ts = np.linspace(0, 0.7, 20001)
        tot = 0.0
        for ax in range(3):
            c = np.array(pr.c[ax][:])
            mono = [c[5], c[4], c[3] / 2, c[2] / 6, c[1] / 24, c[0] / 120]
            q = sum(mono[i] * factorial(i) / factorial(i - d) * ts ** (i - d) for i in range(d, 6))
            tot += np.trapezoid(q * q, ts)
        assert abs(L.orc_primitive_J(C.byref(pr), ctrl) - tot) < 1e-6 * max(1, abs(tot))


def test_waypoint_key_quantisation():
    w = orc.waypoint((0.014, -0.015, 1.005), (0.25, -0.25, 0.04), (0.15, 0, 0), control=orc.JRK)
    key = (C.c_int32 * 13)()
    n = L.orc_waypoint_key(C.byref(w), key)
    assert n == 9
    # round half away from zero on value/resolution computed in f64
    exp = [round(0.014 / 0.01), int(np.sign(-0.015) * np.floor(abs(-0.015 / 0.01) + 0.5)), int(np.floor(1.005 / 0.01 + 0.5))]
    assert [key[0], key[3], key[6]] == exp
    assert key[1] in (2, 3) and key[1] == int(np.floor(0.25 / 0.1 + 0.5))
    w.control = orc.ACC
    assert L.orc_waypoint_key(C.byref(w), key) == 6


def test_float_to_int_cell_convention():
    P = orc.Planner()
    P.set_map(np.zeros((4, 5, 6), np.int8), (1.0, 2.0, 3.0), 0.1)
    assert P.float_to_int((1.05, 2.05, 3.05)) == (0, 0, 0)      # cell centre (voxel_grid.cpp:205-207)
    assert P.float_to_int((1.0999, 2.1001, 3.25)) == (0, 1, 2)
    # a point exactly on the lower map face: round(0/res - 0.5) = round(-0.5) = -1 (half away from zero)
    assert P.float_to_int((0.95, 2.0, 3.0)) == (-1, -1, -1)
    assert P.float_to_int((0.95, 2.0001, 3.0001)) == (-1, 0, 0)
    assert not P.is_free_point((0.95, 2.01, 3.01))
    assert P.is_free_point((1.55, 2.45, 3.35))
    assert not P.is_free_point((1.65, 2.45, 3.35))              # x index 6 >= dim 6


def test_poly_roots_match_numpy():
    rng = np.random.default_rng(5)
    for deg in (1, 2, 3, 4, 5, 6):
        for _ in range(60):
            a = rng.uniform(-3, 3, deg + 1)
            if rng.uniform() < 0.3:
                a[rng.integers(0, deg)] = 0.0
            roots = (C.c_double * 8)()
            n = L.orc_poly_roots_above((C.c_double * (deg + 1))(*a), deg, 0.0, roots)
            got = np.array(roots[:n])
            ref = np.roots(a[::-1])
            ref = np.sort(ref[(abs(ref.imag) < 1e-9) & (ref.real > 1e-9)].real)
            # every reported root is a root; every simple real root is found
            for r in got:
                assert abs(np.polyval(a[::-1], r)) < 1e-7 * (1 + abs(r)) ** deg
            dp = np.polyder(a[::-1])
            simple = [r for r in ref if abs(np.polyval(dp, r)) > 1e-4]
            for r in simple:
                assert np.min(np.abs(got - r)) < 1e-6 * (1 + abs(r)), (a, got, ref)


def _opt_cost(k, T, x0, x1):
    n = 2 * k
    Q = np.zeros((n, n))
    for i in range(k, n):
        for j in range(k, n):
            Q[i, j] = factorial(i) / factorial(i - k) * factorial(j) / factorial(j - k) * T ** (i + j - 2 * k + 1) / (i + j - 2 * k + 1)
    A, b = [], []
    for d, v in enumerate(x0):
        row = np.zeros(n)
        row[d] = factorial(d)
        A.append(row)
        b.append(v)
    for d, v in enumerate(x1):
        if v is None:
            continue
        row = np.zeros(n)
        for i in range(d, n):
            row[i] = factorial(i) / factorial(i - d) * T ** (i - d)
        A.append(row)
        b.append(v)
    A, b = np.array(A), np.array(b)
    m = len(b)
    K = np.block([[2 * Q, A.T], [A, np.zeros((m, m))]])
    x = np.linalg.lstsq(K, np.concatenate([np.zeros(n), b]), rcond=None)[0][:n]
    return x @ Q @ x


@pytest.mark.parametrize("name,sc,gc,k,gfree", [
    ("JRK->JRK", orc.JRK, orc.JRK, 3, [False, False, False]),
    ("JRK->ACC", orc.JRK, orc.ACC, 3, [False, False, True]),
    ("JRK->VEL", orc.JRK, orc.VEL, 3, [False, True, True]),
    ("ACC->ACC", orc.ACC, orc.ACC, 2, [False, False]),
    ("ACC->VEL", orc.ACC, orc.VEL, 2, [False, True])])
def test_heuristic_is_the_optimal_control_cost(name, sc, gc, k, gfree):
    """h(s) = min_T>=t_bar  (min effort to reach the goal set in time T) + w T, solved numerically
    as an equality-constrained QP over polynomial trajectories -- independent of the closed form."""
    rng = np.random.default_rng(sc * 16 + gc)
    P = orc.Planner()
    P.set_map(np.zeros((4, 4, 4), np.int8), (0, 0, 0), 0.1)
    P.set_config(sc, np.zeros((1, 3)), v_max=2.0, w=10.0)
    for _ in range(3):
        s = [rng.uniform(-3, 3, 3), rng.uniform(-2, 2, 3), rng.uniform(-1, 1, 3)]
        g = [rng.uniform(-3, 3, 3), rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3)]
        P.set_goal(orc.waypoint(g[0], g[1], g[2], control=gc))
        h = P.heuristic(orc.waypoint(s[0], s[1], s[2], control=sc))
        tbar = np.abs(g[0] - s[0]).max() / 2.0

        def cost(T):
            return 10.0 * T + sum(_opt_cost(k, T, [s[d][ax] for d in range(k)],
                                            [None if gfree[d] else g[d][ax] for d in range(k)]) for ax in range(3))
        Ts = np.concatenate([[tbar], np.geomspace(max(tbar, 1e-2), max(tbar, 1e-2) * 50 + 20, 600)])
        Ts = Ts[Ts >= tbar]
        cs = np.array([cost(T) for T in Ts])
        i = int(cs.argmin())
        lo, hi = Ts[max(i - 1, 0)], Ts[min(i + 1, len(Ts) - 1)]
        for _ in range(60):
            m1, m2 = lo + (hi - lo) / 3, hi - (hi - lo) / 3
            if cost(m1) < cost(m2):
                hi = m2
            else:
                lo = m1
        ref = min(cs.min(), cost((lo + hi) / 2))
        assert abs(h - ref) < 1e-7 * abs(ref), name


def test_get_succ_control_flow():
    """Pattern of env_poly_map.h:45-69 / env_cloud.h:50-70: successors in control order, tn == curr
    skipped, blocked primitives emitted with +inf cost, t advanced by dt."""
    grid = np.zeros((20, 20, 20), np.int8)
    grid[:, :, 12] = 100  # wall at x index 12
    P = orc.Planner()
    P.set_map(grid, (0, 0, 0), 0.1)
    U = mapgen.control_lattice(1.0, 1, True)
    P.set_config(orc.ACC, U, v_max=2.0, a_max=1.0)
    cur = orc.waypoint((1.05, 1.05, 1.05), control=orc.ACC, t=3.0)
    succ, cost, act = P.get_succ(cur)
    assert len(succ) == 26 and 13 not in act          # u = 0 from rest reproduces curr -> skipped
    assert list(act) == sorted(act)
    for s, c, a in zip(succ, cost, act):
        assert s.t == 4.0
        u = U[a]
        assert np.allclose(s.pos[:], np.array([1.05] * 3) + 0.5 * u) and np.allclose(s.vel[:], u)
        blocked = s.pos[0] > 1.2 - 1e-9 and u[0] > 0   # crosses into the wall cell x=12 (1.2..1.3)
        assert np.isinf(c) == bool(u[0] > 0 and 1.05 + 0.5 * u[0] >= 1.2)
        if not np.isinf(c):
            assert c == float(u @ u) + 10.0            # J(ACC) + w dt
    c = P.counters()
    assert c["n_expansions"] == 1 and c["n_primitives"] == 27 and c["n_succ"] == 26


def test_heuristic_with_unlimited_velocity():
    """v_max <= 0 is the "unlimited" default of the setters: the heuristic must stay a non-negative lower
    bound (no division by a non-positive v_max) and an unbounded-velocity search must still reach the goal."""
    from mpl_ros_amd import mapgen
    grid = np.zeros((32, 32, 32), dtype=np.int8)
    U = mapgen.control_lattice(1.0, 1, True)
    for hid in (False, True):
        P = orc.Planner()
        P.set_map(grid, (0, 0, 0), 0.1)
        P.set_config(orc.ACC, U, v_max=-1.0, a_max=-1.0, heur_ignore_dynamics=hid)
        goal = orc.waypoint((2.55, 2.05, 1.55))
        P.set_goal(goal)
        for pos in ((0.55, 0.55, 0.55), (2.55, 2.05, 0.55), (1.05, 2.95, 1.55)):
            h = P.heuristic(orc.waypoint(pos, vel=(0.5, 0, 0)))
            assert np.isfinite(h) and h > 0
        assert P.heuristic(orc.waypoint((0.55, 0.55, 0.55))) > P.heuristic(orc.waypoint((2.05, 2.05, 1.55)))  # grows with distance
        if hid:  # w * |dp|_inf
            assert P.heuristic(orc.waypoint((0.55, 0.55, 0.55))) == 10.0 * (2.55 - 0.55)
        assert P.plan(orc.waypoint((0.55, 0.55, 0.55)), goal) == orc.OK
        assert P.counters()["n_expansions"] < 2000


def test_blocked_successors_are_accounted_like_upstream_hm():
    """A successor with cost inf gets an hm_ entry and an inf-cost pred entry upstream (env_poly_map.h:60-66
    emits it; GraphSearch stores it).  The oracle keeps them in a side list: every blocked edge re-derives to a
    blocked primitive, and num_states_all >= num_nodes counts the states only blocked primitives reach."""
    from mpl_ros_amd import mapgen
    from tests import util
    grid, origin, res = util.small_map(48, seed=3, occupancy=0.12)
    mapgen.carve_bubble(grid, (1.05, 1.05, 1.05), origin, res, 3)
    U = mapgen.control_lattice(1.0, 1, True)
    P = util.make_oracle(grid, origin, res, orc.ACC, U, v_max=2.0, a_max=1.0, max_expand=800)
    P.plan(orc.waypoint((1.05, 1.05, 1.05)), orc.waypoint((3.55, 3.55, 3.05)))
    c = P.counters()
    par, act = P.blocked_edges()
    assert len(par) == c["n_succ"] - c["n_succ_finite"] > 0
    L = orc.lib()
    for p_id, a in list(zip(par, act))[:200]:
        w, _, _, closed = P.node(int(p_id))
        assert closed
        pr = orc.Primitive()
        u = (orc.C.c_double * 3)(*U[a])
        L.orc_primitive_build(orc.C.byref(w), u, 1.0, orc.C.byref(pr))
        assert not L.orc_is_free_primitive(P.h, orc.C.byref(pr))
    assert P.num_states_all() > P.num_nodes()
    assert P.num_states_all() <= P.num_nodes() + len(par)


def _q1_counts(n, jrk, cap):
    import ctypes as C
    from mpl_ros_amd import mapgen
    from tests import util
    L = orc.lib()
    L.orc_q1_audit.argtypes = [C.c_int]
    L.orc_q1_counts.argtypes = [C.POINTER(C.c_uint64)]
    grid, origin, res, s, g, _ = mapgen.benchmark_map(n)
    U = mapgen.control_lattice(1.0, 2 if jrk else 1, True)
    control = orc.JRK if jrk else orc.ACC
    kw = dict(v_max=2.0, a_max=1.0, tol_pos=0.5, max_expand=cap)
    if jrk:
        kw["j_max"] = 1.0
    P = util.make_oracle(grid, origin, res, control, U, **kw)
    L.orc_q1_audit(1)
    try:
        P.plan(orc.waypoint(s, control=control), orc.waypoint(g, control=control))
        out = (C.c_uint64 * 3)()
        L.orc_q1_counts(out)
    finally:
        L.orc_q1_audit(0)
    return P.counters()["n_expansions"], int(out[0]), int(out[1]), int(out[2])


def test_open_question_q1_pow_versus_multiplication_changes_no_cell():
    """mpl_oracle.h Q1: upstream is recalled to raise t to its powers >= 3 with std::pow, this restatement and the
    device multiply.  Decidable for the searches themselves: with the audit on, every collision sample is evaluated both ways.
    ACC primitives have no cubic term -- not one position bit differs (BASELINE C1, C2, C4-ACC do not depend on Q1 at all).
    JRK primitives: about one sample position in 150 differs in its last bits, and NO sample lands in another cell (29.7 M samples of
    the 256^3 search at 60 000 expansions, 250 000 expansions of the C3 query on the 512^3 map: DESIGN.md 6); asserted here on a
    smaller search."""
    ne, samples, bits, cells = _q1_counts(128, False, 50000)
    assert ne > 500 and samples > 100000 and bits == 0 and cells == 0
    ne, samples, bits, cells = _q1_counts(256, True, 15000)
    assert ne == 15000 and samples > 5_000_000
    assert bits > 0          # (the question is real: the two forms are not the same function ...)
    assert cells == 0        # (... and it changes no voxel index of this search)


def test_open_question_q2_mask_normalisation_at_the_reference_radii():
    """mpl_oracle.h Q2: potential mask distance normalised by the metric radius (here) or by the integer cell radius ceil(r / res)
    (a reviewer's recollection of upstream).  Counted, not argued: at the reference's own setting -- setPotentialRadius(Vec2f(1.5, 1.5))
    on a 0.1 m map, distance_map_planner_node.cpp:187 -- ceil(r / res) = 15 = r / res and the two forms agree except where
    100 (1 - d) sits on an integer up to rounding: 4 of 681 mask values differ, by one unit of potential.  With a radius that is not a
    multiple of the resolution the cell radius itself changes and the masks differ wholesale -- the case the restatement is NOT
    pinned for."""
    import ctypes as C
    L = orc.lib()
    L.orc_q2_mask_audit.argtypes = [C.c_double, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_uint64)]

    def audit(res, radius, pw=1):
        out = (C.c_uint64 * 3)()
        L.orc_q2_mask_audit(res, (C.c_double * 3)(*radius), pw, out)
        return int(out[0]), int(out[1]), int(out[2])
    assert audit(0.1, (1.5, 1.5, 0.0)) == (681, 681, 4)
    assert audit(0.25, (1.0, 1.0, 0.0)) == (45, 45, 0)
    mine, alt, diff = audit(0.1, (1.0, 1.0, 1.0))
    assert mine == alt == 4067 and diff == 60          # (3-D, radius a multiple of the resolution: 1.5 % of the values, one unit each)
    mine, alt, diff = audit(0.1, (0.45, 0.45, 0.0))    # (not a multiple: ceil -> 5 cells instead of 4.5)
    assert diff > mine // 2
