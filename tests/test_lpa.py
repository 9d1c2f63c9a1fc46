"""LPA* incremental replanning: the reference's replanner scenario replayed.

mpl_test_node/src/map_replanner_node.cpp on the `simple` map (launch/map_replanner_node/test.launch): planner_ (A*) and
replan_planner_ (setLPAstar(true)) plan start (14.5, 2.4, 0.025) v (0, 1, 0) -> goal (4, 16, 0.025) with the 2-D lattice,
setTol(0.5, 1, 1); add_cloud.sh fills the columns around a ray (addCloudCallback :206-241 -> updateBlockedNodes),
clear_cloud.sh clears the occupied cells of a ray (clearCloudCallback :175-204 -> updateClearedNodes), replan.sh plans
both again, subtree.sh re-roots the LPA* state space one primitive ahead (subtreeCallback :243-255).

CPU: the oracle's LPA* (UNVERIFIED restatement, oracle/mpl_oracle_lpa.inc) against its own fresh A*: same cost after
every edit, fewer expansions.  GPU: the HIP LPA* against the oracle's, bit for bit -- expansion order, the whole state
space (g, rhs, flags, predecessor entries with their blocked flags), cost, trajectory."""
import os

import numpy as np
import pytest

from mpl_ros_amd import mapgen
from oracle import orc
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
START, START_V, GOAL = (14.5, 2.4, 0.025), (0.0, 1.0, 0.0), (4.0, 16.0, 0.025)
KW = dict(v_max=2.0, a_max=1.0, tol_pos=0.5, tol_vel=1.0, tol_acc=1.0)


def simple_map():
    d = np.load(os.path.join(ROOT, "tests", "golden", "simple_map.npz"))
    return d["grid"].copy(), tuple(d["origin"].tolist()), float(d["res"])


def add_cloud_cells(P, p1, p2):
    """addCloudCallback: the free cells of the 5 x 5 neighbourhoods along the ray (their whole columns get filled)"""
    new = []
    for pn in P.ray_trace(p1, p2):
        for nx in range(-2, 3):
            for ny in range(-2, 3):
                c = (int(pn[0]) + nx, int(pn[1]) + ny, int(pn[2]))
                if P.cell_state(c) == 0:
                    new.append(c)
    return new


def clear_cloud_cells(P, p1, p2):
    """clearCloudCallback: the occupied cells of the ray (their whole columns get cleared)"""
    return [tuple(int(v) for v in pn) for pn in P.ray_trace(p1, p2) if P.cell_state(pn) == 1]


def edit_columns(grid, cells, value):
    for x, y, _ in cells:
        grid[:, y, x] = value


class Scenario:
    """The map edits of the scripts, as (kind, cells) steps computed on a scratch oracle map."""

    def __init__(self):
        self.grid, self.origin, self.res = simple_map()
        self.U = mapgen.control_lattice(1.0, 1, False)
        self.scratch = util.make_oracle(self.grid, self.origin, self.res, orc.ACC, self.U, **KW)

    def add(self, p1, p2):
        cells = add_cloud_cells(self.scratch, p1, p2)
        edit_columns(self.grid, cells, 100)
        self.scratch.set_map(self.grid, self.origin, self.res)
        return cells

    def clear(self, p1, p2):
        cells = clear_cloud_cells(self.scratch, p1, p2)
        edit_columns(self.grid, cells, 0)
        self.scratch.set_map(self.grid, self.origin, self.res)
        return cells

    def clear_cells(self, cells):
        edit_columns(self.grid, cells, 0)
        self.scratch.set_map(self.grid, self.origin, self.res)
        return cells


def oracle_pair(sc):
    A = util.make_oracle(sc.grid, sc.origin, sc.res, orc.ACC, sc.U, **KW)
    L = util.make_oracle(sc.grid, sc.origin, sc.res, orc.ACC, sc.U, **KW)
    L.set_lpastar(True)
    return A, L


def test_oracle_lpastar_equals_fresh_astar_on_the_replanner_scenario():
    sc = Scenario()
    A, L = oracle_pair(sc)
    start, goal = orc.waypoint(START, vel=START_V), orc.waypoint(GOAL)

    def both(start):
        sa = A.plan(start, goal)
        sl = L.plan(start, goal)
        assert sa == sl == orc.OK and A.traj_cost == L.traj_cost
        return len(A.expanded()[0]), L.lpa_iterations()

    a0, l0 = both(start)
    assert a0 == l0 and L.initialized()  # the first LPA* plan IS an A*
    cost0 = L.traj_cost
    # add_cloud.sh
    new_obs = sc.add((12.55, 9.55, 0.025), (12.55, 11.05, 0.025))
    assert len(new_obs) > 100
    for P in (A, L):
        P.set_map(sc.grid, sc.origin, sc.res)
    assert L.update_blocked(new_obs) > 0
    a1, l1 = both(start)
    assert L.traj_cost > cost0 and 0 < l1 < a1  # the obstacle sits on the old path; the repair expands less than a fresh A*
    # clear_cloud.sh (a partial clearing: no primitive gets through yet), then the whole obstacle is removed
    cl = sc.clear((12.75, 9.55, 0.025), (12.65, 11.95, 0.025))
    for P in (A, L):
        P.set_map(sc.grid, sc.origin, sc.res)
    L.update_cleared(cl)
    both(start)
    rest = [c for c in new_obs if sc.scratch.cell_state(c) == 1]
    sc.clear_cells(rest)
    for P in (A, L):
        P.set_map(sc.grid, sc.origin, sc.res)
    assert L.update_cleared(rest) > 0
    a3, l3 = both(start)
    assert L.traj_cost == cost0 and l3 < a3  # back on the original optimum
    # subtree.sh + replan.sh: move one primitive ahead
    tr = L.traj()
    first_cost = L.traj_cost
    L.sub_state_space(1)
    w1 = tr["wps"][1]
    start1 = orc.waypoint(tuple(w1.pos), vel=tuple(w1.vel))
    a4, l4 = both(start1)
    assert l4 < a4 and L.traj_cost < first_cost


def gpu_pair(sc):
    from mpl_ros_amd.planner import VoxelMapPlanner
    mu, a = util.make_gpu(sc.grid, sc.origin, sc.res, sc.U, **KW)
    l = VoxelMapPlanner(False)
    l.setMapUtil(mu)  # the two planners share one MapUtil, like planner_ / replan_planner_
    l.setVmax(2.0); l.setAmax(1.0); l.setDt(1.0); l.setU(sc.U); l.setTol(0.5, 1, 1)
    l.setCapacity(1, 1 << 17, 1 << 19, 1 << 19)
    l.setLPAstar(True)
    return mu, a, l


def set_gpu_map(mu, sc):
    dz, dy, dx = sc.grid.shape
    mu.setMap(sc.origin, (dx, dy, dz), sc.grid.ravel(), sc.res)  # setMap(map_util, map), map_replanner_node.cpp:188,226


def compare_lpa(L, l, r, st_o):
    """HIP LPA* planner `l` (result r) against the oracle's `L` after the same call sequence: bit-exact"""
    assert r.status == st_o
    ids = L.expanded()[0]
    assert r.n_expanded == L.lpa_iterations() == len(ids) and r.expand_hash == util.expand_hash(ids)
    ss = l.lpaStateSpace()
    n = L.num_nodes()
    assert ss["n_nodes"] == n == r.n_nodes
    g = np.array([L.node(i)[1] for i in range(n)]); h = np.array([L.node(i)[2] for i in range(n)])
    closed = np.array([L.node(i)[3] for i in range(n)], dtype=np.int32)
    rhs = np.array([L.node_rhs(i) for i in range(n)]); opened = np.array([L.node_opened(i) for i in range(n)], dtype=np.int32)
    assert np.array_equal(ss["g"], g) and np.array_equal(ss["rhs"], rhs) and np.array_equal(ss["h"], h)
    assert np.array_equal(ss["closed"], closed) and np.array_equal(ss["opened"], opened)
    co, po, ao = L.edges()
    assert np.array_equal(ss["child"], co) and np.array_equal(ss["parent"], po) and np.array_equal(ss["action"], ao)
    assert np.array_equal(ss["blocked"], L.edges_blocked())
    assert r.n_closed == L.num_closed()
    if st_o == orc.OK:
        assert r.cost == L.traj_cost
        to, tg = L.traj(), l.getTraj()
        assert np.array_equal(tg.actions, to["actions"]) and np.array_equal(tg.node_ids, to["node_ids"])
        for wg, wo in zip(tg.getWaypoints(), to["wps"]):
            assert np.array_equal(wg.state(), orc.wp_state(wo, orc.ACC))


@pytest.mark.gpu
def test_hip_lpastar_replays_the_replanner_scenario_bit_exact():
    sc = Scenario()
    A, L = oracle_pair(sc)
    mu, a, l = gpu_pair(sc)
    start_o, goal_o = orc.waypoint(START, vel=START_V), orc.waypoint(GOAL)
    start_g, goal_g = util.gpu_wp(START, vel=START_V), util.gpu_wp(GOAL)

    def replan(start_o, start_g):
        """replanCallback: planner_.plan then replan_planner_.plan, on both sides"""
        sa = A.plan(start_o, goal_o)
        ok_a = a.plan(start_g, goal_g)
        ra = a.getResult()
        assert ok_a == (sa == orc.OK) and ra.n_expanded == len(A.expanded()[0]) and ra.cost == A.traj_cost
        L.reset_counters()
        sl = L.plan(start_o, goal_o)
        ok_l = l.plan(start_g, goal_g)
        rl = l.getResult()
        assert ok_l == (sl == orc.OK)
        compare_lpa(L, l, rl, sl)
        assert rl.cost == ra.cost  # LPA* cost == fresh A* cost
        return ra, rl

    ra0, rl0 = replan(start_o, start_g)
    assert l.initialized() and rl0.n_expanded == ra0.n_expanded
    cost0 = rl0.cost
    # add_cloud.sh
    new_obs = sc.add((12.55, 9.55, 0.025), (12.55, 11.05, 0.025))
    for P in (A, L):
        P.set_map(sc.grid, sc.origin, sc.res)
    set_gpu_map(mu, sc)
    nb_o = L.update_blocked(new_obs)
    assert l.updateBlockedNodes(new_obs) == nb_o > 0
    ra1, rl1 = replan(start_o, start_g)
    assert rl1.cost > cost0 and 0 < rl1.n_expanded < ra1.n_expanded
    # clear_cloud.sh, then the rest of the obstacle
    cl = sc.clear((12.75, 9.55, 0.025), (12.65, 11.95, 0.025))
    for P in (A, L):
        P.set_map(sc.grid, sc.origin, sc.res)
    set_gpu_map(mu, sc)
    assert l.updateClearedNodes(cl) == L.update_cleared(cl)
    replan(start_o, start_g)
    rest = [c for c in new_obs if sc.scratch.cell_state(c) == 1]
    sc.clear_cells(rest)
    for P in (A, L):
        P.set_map(sc.grid, sc.origin, sc.res)
    set_gpu_map(mu, sc)
    nc_o = L.update_cleared(rest)
    assert l.updateClearedNodes(rest) == nc_o > 0
    ra3, rl3 = replan(start_o, start_g)
    assert rl3.cost == cost0 and rl3.n_expanded < ra3.n_expanded
    # subtree.sh: getSubStateSpace(1), start = getTraj().getWaypoints()[1]; replan.sh
    tg = l.getTraj()
    L.sub_state_space(1)
    l.getSubStateSpace(1)
    w1 = tg.getWaypoints()[1]
    assert np.array_equal(w1.pos, np.array(L.traj()["wps"][1].pos[:]))
    ra4, rl4 = replan(orc.waypoint(tuple(w1.pos), vel=tuple(w1.vel)), util.gpu_wp(tuple(w1.pos), vel=tuple(w1.vel)))
    assert rl4.n_expanded < ra4.n_expanded and rl4.cost < cost0


@pytest.mark.gpu
def test_lpastar_replanning_cycles_do_not_exhaust_the_pools():
    """Ten cycles of plan -> obstacle on the path -> repair -> obstacle removed -> repair -> move one primitive ahead on
    pools that hold four times the first plan's state space: getSubStateSpace rebuilds into the second pool set, which
    compacts what the edits and repairs accumulate.  Every plan's cost equals a fresh A*'s on the same map."""
    sc = Scenario()
    mu, a, l = gpu_pair(sc)
    goal_g = util.gpu_wp(GOAL)
    start_g = util.gpu_wp(START, vel=START_V)
    assert l.plan(start_g, goal_g)
    first = l.lpaStateSpace()
    l.setCapacity(1, 4 * first["n_nodes"], 4 * first["n_edges"], 8 * first["n_edges"])
    cycles = 0
    for k in range(10):
        assert l.plan(start_g, goal_g) and a.plan(start_g, goal_g)
        assert l.getResult().cost == a.getResult().cost
        tr = l.getTraj()
        if len(tr.segs) < 4:
            break
        # an obstacle two primitives ahead on the current path, then removed again
        w = tr.getWaypoints()[2]
        cells = [c for c in add_cloud_cells(sc.scratch, (w.pos[0] - 0.1, w.pos[1], 0.025), (w.pos[0] + 0.1, w.pos[1], 0.025))]
        if cells:
            edit_columns(sc.grid, cells, 100)
            sc.scratch.set_map(sc.grid, sc.origin, sc.res)
            set_gpu_map(mu, sc)
            l.updateBlockedNodes(cells)
            ok_l, ok_a = l.plan(start_g, goal_g), a.plan(start_g, goal_g)
            assert ok_l == ok_a and (not ok_l or l.getResult().cost == a.getResult().cost)
            sc.clear_cells(cells)
            set_gpu_map(mu, sc)
            l.updateClearedNodes(cells)
            assert l.plan(start_g, goal_g) and a.plan(start_g, goal_g) and l.getResult().cost == a.getResult().cost
        tr = l.getTraj()
        l.getSubStateSpace(1)
        w1 = tr.getWaypoints()[1]
        start_g = util.gpu_wp(tuple(w1.pos), vel=tuple(w1.vel))
        cycles += 1
    assert cycles >= 5
    ss = l.lpaStateSpace()
    assert ss["n_nodes"] <= 4 * first["n_nodes"] and ss["n_edges"] <= 4 * first["n_edges"]


# ---------------------------------------------------------------- round 4: LPA* on 3-D voxel maps
# The scenario of map_replanner_node.cpp:175-255 on a 3-D map with the 27-input lattice: plan; an obstacle (a cube of
# voxels: what addCloudCallback's inflated cloud amounts to in 3-D) lands on the middle of the current trajectory
# (updateBlockedNodes); it is removed again (updateClearedNodes); the robot moves one primitive ahead (getSubStateSpace(1)).
KW3 = dict(v_max=2.0, a_max=1.0, tol_pos=0.5)


def box_cells(P, center, half):
    """free cells of the (2 half + 1)^3 cube around the cell of `center` (P: a scratch oracle holding the current map)"""
    c = P.float_to_int(center)
    out = []
    for dz in range(-half, half + 1):
        for dy in range(-half, half + 1):
            for dx in range(-half, half + 1):
                cc = (c[0] + dx, c[1] + dy, c[2] + dz)
                if P.cell_state(cc) == 0:
                    out.append(cc)
    return out


def scenario_3d(name):
    if name == "skir":  # the reference's one surviving 3-D map and its launch query (launch/map_planner_node/test.launch.skir)
        d = np.load(os.path.join(ROOT, "tests", "golden", "skir_map.npz"))
        return d["grid"].copy(), tuple(d["origin"].tolist()), float(d["res"]), (5.5, 5.5, 0.5), (1.0, 0.0, 0.0), (1.5, 1.5, 5.5)
    grid, origin, res, start, goal, _ = mapgen.benchmark_map(256)  # BASELINE C2: 256^3 random boxes, 31 309 expansions to the goal
    return grid, origin, res, start, (0.0, 0.0, 0.0), goal


def test_oracle_lpastar_equals_fresh_astar_on_a_3d_map():
    grid, origin, res, start, sv, goal = scenario_3d("skir")
    U = mapgen.control_lattice(1.0, 1, True)
    A = util.make_oracle(grid, origin, res, orc.ACC, U, **KW3)
    L = util.make_oracle(grid, origin, res, orc.ACC, U, **KW3)
    L.set_lpastar(True)
    so, go = orc.waypoint(start, vel=sv), orc.waypoint(goal)
    assert A.plan(so, go) == L.plan(so, go) == orc.OK and A.traj_cost == L.traj_cost
    cost0, n0 = L.traj_cost, L.lpa_iterations()
    tr = L.traj()
    cells = box_cells(A, tuple(tr["wps"][tr["n"] // 2].pos), 2)
    g2 = grid.copy()
    for x, y, z in cells:
        g2[z, y, x] = 100
    for P in (A, L):
        P.set_map(g2, origin, res)
    assert L.update_blocked(cells) > 0
    L.reset_counters()
    assert A.plan(so, go) == L.plan(so, go) == orc.OK and A.traj_cost == L.traj_cost > cost0
    assert 0 < L.lpa_iterations() < len(A.expanded()[0])
    for P in (A, L):
        P.set_map(grid, origin, res)
    assert L.update_cleared(cells) > 0
    L.reset_counters()
    assert A.plan(so, go) == L.plan(so, go) == orc.OK and A.traj_cost == L.traj_cost == cost0
    assert L.lpa_iterations() < n0


def test_oracle_subspace_by_fresh_plan_keeps_cost_and_the_stored_trajectory():
    """L5b (round 6): getSubStateSpace realised as a fresh plan from the k-th path state.  Same path cost as the Dijkstra
    realisation (L5) and as a fresh A* from the new start; the stored trajectory survives; the next plan() from the new root
    finds a consistent space (no expansion)."""
    grid, origin, res, start, sv, goal = scenario_3d("skir")
    U = mapgen.control_lattice(1.0, 1, True)
    res_by_mode = {}
    for mode in (0, 1):
        L = util.make_oracle(grid, origin, res, orc.ACC, U, **KW3)
        L.set_lpastar(True)
        L.set_reroot(mode)
        so, go = orc.waypoint(start, vel=sv), orc.waypoint(goal)
        assert L.plan(so, go) == orc.OK
        tr = L.traj()
        L.sub_state_space(1)
        tr2 = L.traj()
        assert np.array_equal(tr["actions"], tr2["actions"]) and tr2["node_ids"][1] == 0  # the new root is state 0 of the new space
        w1 = tr["wps"][1]
        s1 = orc.waypoint(tuple(w1.pos), vel=tuple(w1.vel))
        L.reset_counters()
        assert L.plan(s1, go) == orc.OK
        res_by_mode[mode] = (L.traj_cost, L.lpa_iterations(), L.num_nodes())
    A = util.make_oracle(grid, origin, res, orc.ACC, U, **KW3)
    assert A.plan(s1, go) == orc.OK
    assert res_by_mode[0][0] == res_by_mode[1][0] == A.traj_cost
    assert res_by_mode[1][1] == 0 and res_by_mode[1][2] <= res_by_mode[0][2]  # consistent already; the fresh space is the smaller one


@pytest.mark.gpu
@pytest.mark.parametrize("name,reroot", [("skir", 0), ("skir", 1), ("c2_256", 0), ("c2_256", 2)])
def test_hip_lpastar_on_3d_maps_bit_exact_and_equal_to_fresh_astar(name, reroot):
    """LPA* replayed on the skir 3-D map and on the 256^3 random-box map of BASELINE C2 (27-input lattice): the HIP LPA*
    against the oracle's bit for bit (expansion order, every state's g / rhs / flags, every predecessor entry with its
    blocked flag, trajectory), and after every repair cost == a fresh device A* on the edited map.  Prints the kernel time
    of the repair next to the fresh A*'s."""
    from mpl_ros_amd.planner import VoxelMapPlanner
    grid, origin, res, start, sv, goal = scenario_3d(name)
    U = mapgen.control_lattice(1.0, 1, True)
    scratch = util.make_oracle(grid, origin, res, orc.ACC, U, **KW3)
    L = util.make_oracle(grid, origin, res, orc.ACC, U, **KW3)
    L.set_lpastar(True)
    big = name != "skir"
    caps = dict(max_nodes=1 << 18, max_edges=1 << 20, max_log=1 << 20) if big else {}
    mu, a = util.make_gpu(grid, origin, res, U, **KW3, **caps)
    l = VoxelMapPlanner(False)
    l.setMapUtil(mu)
    l.setVmax(2.0); l.setAmax(1.0); l.setDt(1.0); l.setU(U); l.setTol(0.5)
    l.setCapacity(1, 1 << 18, 1 << 20, 1 << 21) if big else l.setCapacity(1, 1 << 16, 1 << 18, 1 << 18)
    l.setLPAstar(True)
    l.setSubStateSpaceMode(reroot)  # (0 Dijkstra, 1 fresh plan, 2 auto: the fresh plan at C2 size)
    L.set_reroot(reroot)
    go, gg = orc.waypoint(goal), util.gpu_wp(goal)
    times = []

    def replan(label, so, sg):
        L.reset_counters()
        sl = L.plan(so, go)
        ok_l = l.plan(sg, gg)
        rl = l.getResult()
        lpa_ms = l.lastKernelMs()
        assert ok_l == (sl == orc.OK)
        compare_lpa(L, l, rl, sl)
        ok_a = a.plan(sg, gg)
        ra = a.getResult()
        assert ok_a == ok_l and rl.cost == ra.cost  # LPA* cost == fresh A* cost on the same (edited) map
        times.append((label, int(rl.n_expanded), round(lpa_ms, 3), int(ra.n_expanded), round(a.lastKernelMs(), 3)))
        return rl, ra

    def set_maps(g):
        scratch.set_map(g, origin, res)
        L.set_map(g, origin, res)
        dz, dy, dx = g.shape
        mu.setMap(origin, (dx, dy, dz), g.ravel(), res)

    so, sg = orc.waypoint(start, vel=sv), util.gpu_wp(start, vel=sv)
    rl0, ra0 = replan("first plan", so, sg)
    assert l.initialized() and rl0.n_expanded == ra0.n_expanded
    cost0 = rl0.cost
    tr = L.traj()
    cells = box_cells(scratch, tuple(tr["wps"][tr["n"] // 2].pos), 2)
    g2 = grid.copy()
    for x, y, z in cells:
        g2[z, y, x] = 100
    set_maps(g2)
    nb = L.update_blocked(cells)
    assert l.updateBlockedNodes(cells) == nb > 0
    rl1, ra1 = replan("box on the path", so, sg)
    assert rl1.cost >= cost0 and 0 < rl1.n_expanded < ra1.n_expanded
    set_maps(grid)
    nc = L.update_cleared(cells)
    assert l.updateClearedNodes(cells) == nc > 0
    rl2, ra2 = replan("box removed", so, sg)
    assert rl2.cost == cost0 and rl2.n_expanded < ra2.n_expanded
    tg = l.getTraj()
    L.sub_state_space(1)
    l.getSubStateSpace(1)
    w1 = tg.getWaypoints()[1]
    rl3, ra3 = replan("one primitive ahead", orc.waypoint(tuple(w1.pos), vel=tuple(w1.vel)), util.gpu_wp(tuple(w1.pos), vel=tuple(w1.vel)))
    assert rl3.n_expanded < ra3.n_expanded and rl3.cost < cost0
    if reroot == 1 or (reroot == 2 and big):
        assert rl3.n_expanded == 0  # re-rooted by a fresh plan: the space is consistent for the new start
    print(f"LPA* on {name}: (step, LPA* expansions, LPA* kernel ms, fresh A* expansions, fresh A* kernel ms)", times)


# ---------------------------------------------------------------- round 4: the goal is a region (choice L7)
# tests/golden/lpa_goal_region_case.npz: a 40 x 40 cell world found by a random search for "LPA* != fresh A*" (5951 replans
# agree since).  Four single-cell edits near the goal: a cheaper state G1 of the goal region appears, then its own cell is
# occupied.  Following only the last plan's goal state, the search ran G1 to g = inf and came back with another state of the
# region at cost 56 although the settled state a fresh A* returns (cost 67, smaller key) was sitting in the pool.
GOAL_REGION_EDITS = [(36, 37, False), (28, 28, True), (27, 27, True), (30, 30, False)]  # (x, y, was occupied)


def _goal_region_case():
    grid = np.load(os.path.join(ROOT, "tests", "golden", "lpa_goal_region_case.npz"))["grid"].copy()
    return grid, (0.0, 0.0, 0.0), 0.25, (1.125, 1.125, 0.125), (8.125, 8.125, 0.125)


def test_oracle_lpastar_follows_the_best_settled_state_of_the_goal_region():
    grid, origin, res, start, goal = _goal_region_case()
    U = mapgen.control_lattice(1.0, 1, False)
    A = util.make_oracle(grid, origin, res, orc.ACC, U, **KW3)
    L = util.make_oracle(grid, origin, res, orc.ACC, U, **KW3)
    L.set_lpastar(True)
    so, go = orc.waypoint(start), orc.waypoint(goal)
    assert A.plan(so, go) == L.plan(so, go) == orc.OK and A.traj_cost == L.traj_cost
    costs = []
    for cx, cy, occ in GOAL_REGION_EDITS:
        grid[0, cy, cx] = 0 if occ else 100
        for P in (A, L):
            P.set_map(grid, origin, res)
        (L.update_cleared if occ else L.update_blocked)([(cx, cy, 0)])
        assert A.plan(so, go) == L.plan(so, go) == orc.OK and A.traj_cost == L.traj_cost
        costs.append(L.traj_cost)
    assert costs == [67.0, 58.0, 56.0, 67.0]


@pytest.mark.gpu
def test_hip_lpastar_goal_region_case_matches_the_oracle_and_a_fresh_astar():
    from mpl_ros_amd.planner import VoxelMapPlanner
    grid, origin, res, start, goal = _goal_region_case()
    U = mapgen.control_lattice(1.0, 1, False)
    L = util.make_oracle(grid, origin, res, orc.ACC, U, **KW3)
    L.set_lpastar(True)
    mu, a = util.make_gpu(grid, origin, res, U, **KW3)
    l = VoxelMapPlanner(False)
    l.setMapUtil(mu)
    l.setVmax(2.0); l.setAmax(1.0); l.setDt(1.0); l.setU(U); l.setTol(0.5)
    l.setCapacity(1, 1 << 16, 1 << 18, 1 << 18)
    l.setLPAstar(True)
    so, go, sg, gg = orc.waypoint(start), orc.waypoint(goal), util.gpu_wp(start), util.gpu_wp(goal)

    def both():
        L.reset_counters()
        sl = L.plan(so, go)
        assert l.plan(sg, gg) == (sl == orc.OK)
        compare_lpa(L, l, l.getResult(), sl)
        assert a.plan(sg, gg) and a.getResult().cost == l.getResult().cost
        return l.getResult().cost

    costs = [both()]
    for cx, cy, occ in GOAL_REGION_EDITS:
        grid[0, cy, cx] = 0 if occ else 100
        L.set_map(grid, origin, res)
        dz, dy, dx = grid.shape
        mu.setMap(origin, (dx, dy, dz), grid.ravel(), res)
        if occ:
            assert l.updateClearedNodes([(cx, cy, 0)]) == L.update_cleared([(cx, cy, 0)])
        else:
            assert l.updateBlockedNodes([(cx, cy, 0)]) == L.update_blocked([(cx, cy, 0)])
        costs.append(both())
    assert costs == [67.0, 67.0, 58.0, 56.0, 67.0]


@pytest.mark.gpu
def test_hip_lpastar_imports_a_jerk_lattice_plan_bit_exact_and_survives_degenerate_starts():
    """Round 5: a fresh LPA* plan is planned by the speculative kernel and imported into the LPA* pools (mplx_lpa.h).  The other tests
    cover the 27-input ACC lattice; here the 125-input JERK lattice (160-byte records, units of two waves, 128-lane import of the
    blocked log), capped, against the CPU LPA* -- every state's g / rhs / h / flags, every entry, the blocked log -- then a repair on
    the imported space after an obstacle lands on the path; and the two plans that never search (start inside the goal region, start
    occupied) leave no state space behind, like the one-workgroup kernel."""
    from mpl_ros_amd.planner import VoxelMapPlanner
    grid, origin, res = util.small_map(64, seed=21, occupancy=0.06)
    U = mapgen.control_lattice(1.0, 2, True)
    assert U.shape[0] == 125
    kw = dict(v_max=2.0, a_max=1.0, j_max=1.0, tol_pos=0.5, max_expand=4000)
    L = util.make_oracle(grid, origin, res, orc.JRK, U, **kw)
    L.set_lpastar(True)
    mu, a = util.make_gpu(grid, origin, res, U, **kw)
    l = VoxelMapPlanner(False)
    l.setMapUtil(mu)
    l.setVmax(2.0); l.setAmax(1.0); l.setJmax(1.0); l.setDt(1.0); l.setU(U); l.setTol(0.5); l.setMaxNum(4000)
    l.setCapacity(1, 1 << 19, 1 << 22, 1 << 21)
    l.setLPAstar(True)
    start, goal = (0.55, 0.55, 0.55), (5.55, 5.55, 5.55)
    so, go = orc.waypoint(start, control=orc.JRK), orc.waypoint(goal, control=orc.JRK)
    sg, gg = util.gpu_wp(start, control=orc.JRK), util.gpu_wp(goal, control=orc.JRK)
    L.reset_counters()
    st = L.plan(so, go)
    ok = l.plan(sg, gg)
    r = l.getResult()
    assert ok == (st == orc.OK) and r.n_expanded > 500

    def compare(L, l, r, st_o):  # (compare_lpa with JRK waypoints)
        assert r.status == st_o
        ids = L.expanded()[0]
        assert r.n_expanded == L.lpa_iterations() == len(ids) and r.expand_hash == util.expand_hash(ids)
        ss = l.lpaStateSpace()
        n = L.num_nodes()
        assert ss["n_nodes"] == n == r.n_nodes
        g = np.array([L.node(i)[1] for i in range(n)]); h = np.array([L.node(i)[2] for i in range(n)])
        closed = np.array([L.node(i)[3] for i in range(n)], dtype=np.int32)
        rhs = np.array([L.node_rhs(i) for i in range(n)]); opened = np.array([L.node_opened(i) for i in range(n)], dtype=np.int32)
        assert np.array_equal(ss["g"], g) and np.array_equal(ss["rhs"], rhs) and np.array_equal(ss["h"], h)
        assert np.array_equal(ss["closed"], closed) and np.array_equal(ss["opened"], opened)
        co, po, ao = L.edges()
        assert np.array_equal(ss["child"], co) and np.array_equal(ss["parent"], po) and np.array_equal(ss["action"], ao)
        assert np.array_equal(ss["blocked"], L.edges_blocked())
        if st_o == orc.OK:
            assert r.cost == L.traj_cost
    compare(L, l, r, st)
    if st == orc.OK:  # an obstacle on the path, a repair on the imported space
        tr = L.traj()
        cells = box_cells(L, tuple(tr["wps"][tr["n"] // 2].pos), 1)
        g2 = grid.copy()
        for x, y, z in cells:
            g2[z, y, x] = 100
        L.set_map(g2, origin, res)
        dz, dy, dx = g2.shape
        mu.setMap(origin, (dx, dy, dz), g2.ravel(), res)
        assert l.updateBlockedNodes(cells) == L.update_blocked(cells)
        L.reset_counters()
        st2 = L.plan(so, go)
        ok2 = l.plan(sg, gg)
        assert ok2 == (st2 == orc.OK)
        compare(L, l, l.getResult(), st2)
        mu.setMap(origin, (dx, dy, dz), grid.ravel(), res)
    # plans that never search: no state space is left behind
    l2 = VoxelMapPlanner(False)
    l2.setMapUtil(mu)
    l2.setVmax(2.0); l2.setAmax(1.0); l2.setJmax(1.0); l2.setDt(1.0); l2.setU(U); l2.setTol(0.5)
    l2.setCapacity(1, 1 << 19, 1 << 22, 1 << 21)
    l2.setLPAstar(True)
    assert l2.plan(sg, util.gpu_wp((0.75, 0.55, 0.55), control=orc.JRK))  # the start already satisfies the goal
    assert l2.getResult().cost == 0.0 and not l2.initialized()
    occ = np.argwhere(grid > 0)[0]
    p_occ = ((float(occ[2]) + 0.5) * res + origin[0], (float(occ[1]) + 0.5) * res + origin[1], (float(occ[0]) + 0.5) * res + origin[2])  # (grid is z, y, x)
    assert not l2.plan(util.gpu_wp(p_occ, control=orc.JRK), gg)
    assert l2.getResult().status == 2 and not l2.initialized()
