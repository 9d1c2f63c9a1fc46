"""Potential-field cost and search region of the map planner: the reference's distance-map planner flow.

mpl_test_node/src/distance_map_planner_node.cpp on a 2-D slice of the `simple` map: plan (tol 0.2); then around that
path setSearchRadius(0.5, 0.5) + setSearchRegion(path) + setPotentialRadius(1.5, 1.5) + setPotentialWeight(10) +
setGradientWeight(0) + updatePotentialMap(start) and plan again (:185-193); then the potential alone on the whole map
(:208-224); getSearchRegion (:199) and getPotentialCloud (:231).

CPU: properties of the oracle's restatement (UNVERIFIED upstream semantics, oracle/mpl_oracle_pot.inc P1-P3).
GPU: the HIP path against it, bit for bit -- auxiliary map, get_succ costs, whole plans."""
import os

import numpy as np
import pytest

from mpl_ros_amd import mapgen
from oracle import orc
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
START, GOAL = (14.5, 4.5, 0.0), (2.4, 16.6, 0.0)   # launch/distance_map_planner_node/test.launch:16-33
KW = dict(v_max=2.0, a_max=1.0)


def slice_map():
    """distance_map_planner_node.cpp:12-47 slices the voxel map into an occupancy map; here: one layer of the `simple`
    map, whose centre plane is z = 0 (the 2-D planners' convention of mpl_ros_amd.planner.OccMapUtil)."""
    d = np.load(os.path.join(ROOT, "tests", "golden", "simple_map.npz"))
    grid = d["grid"][1:2].copy()
    res = float(d["res"])
    return grid, (float(d["origin"][0]), float(d["origin"][1]), -0.5 * res), res


def oracle(tol):
    grid, origin, res = slice_map()
    return util.make_oracle(grid, origin, res, orc.ACC, mapgen.control_lattice(1.0, 1, False), tol_pos=tol, **KW)


def first_path():
    P = oracle(0.2)
    assert P.plan(orc.waypoint(START), orc.waypoint(GOAL)) == orc.OK
    return P, [tuple(w.pos) for w in P.traj()["wps"]]


def test_oracle_potential_and_region_properties():
    P, path = first_path()
    plain_cost = P.traj_cost
    grid, origin, res = slice_map()
    # P1: the mask is a cone of height 100 around every obstacle, clipped at the radius
    Q = oracle(0.5)
    Q.set_potential_weights(10, 0)
    Q.update_potential_map((1.5, 1.5, 0), START)
    a = Q.aux_map()[0]
    occ = grid[0] > 0
    assert np.array_equal(a == 100, occ) and a.min() == 0
    ys, xs = np.nonzero((a > 0) & (a < 100))
    oy, ox = np.nonzero(occ)
    for y, x in list(zip(ys, xs))[::97]:
        d = np.sqrt(((oy - y) * res / 1.5) ** 2 + ((ox - x) * res / 1.5) ** 2).min()
        assert a[y, x] == int(100.0 * (1.0 - d))
    # P3: with the potential the optimum keeps away from the walls: costlier than the plain optimum, never cheaper
    assert Q.plan(orc.waypoint(START), orc.waypoint(GOAL)) == orc.OK and Q.traj_cost >= plain_cost
    # P2: the region is the 0.5 m box neighbourhood of the joined-up path; every expanded state stays inside it
    R = oracle(0.5)
    R.set_search_region(path, (0.5, 0.5, 0))
    reg = R.aux_map()[0] >= 0
    assert 0 < reg.sum() < reg.size // 4
    for p in path:
        c = R.float_to_int(p)
        assert reg[c[1] - 5:c[1] + 6, c[0] - 5:c[0] + 6].all()
    assert R.plan(orc.waypoint(START), orc.waypoint(GOAL)) == orc.OK and R.traj_cost >= plain_cost
    _, pos = R.expanded()
    for q in pos[::7]:
        c = R.float_to_int(q)
        assert reg[c[1], c[0]]
    # removing the region gives the plain search back
    R.set_search_region([], (0.5, 0.5, 0))
    assert R.plan(orc.waypoint(START), orc.waypoint(GOAL)) == orc.OK and R.traj_cost == oracle_cost(0.5)


def oracle_cost(tol):
    P = oracle(tol)
    assert P.plan(orc.waypoint(START), orc.waypoint(GOAL)) == orc.OK
    return P.traj_cost


def gpu(tol):
    from mpl_ros_amd.planner import OccMapPlanner, OccMapUtil
    grid, origin, res = slice_map()
    mu = OccMapUtil()
    mu.setMap(origin[:2], (grid.shape[2], grid.shape[1]), grid.ravel(), res)
    mu.freeUnknown()
    pl = OccMapPlanner(False)
    pl.setMapUtil(mu)
    pl.setVmax(2.0); pl.setAmax(1.0); pl.setDt(1.0); pl.setEpsilon(1.0)
    pl.setU(mapgen.control_lattice(1.0, 1, False)[:, :2])
    pl.setTol(tol)
    return mu, pl


def wp2(p):
    from mpl_ros_amd.planner import ACC, Waypoint2D
    w = Waypoint2D(ACC)
    w.pos, w.vel = np.array(p[:2], float), np.zeros(2)
    return w


@pytest.mark.gpu
def test_hip_distance_map_planner_flow_matches_the_oracle():
    from mpl_ros_amd import _capi
    import ctypes as C
    P, path = first_path()
    mu, pl = gpu(0.2)
    assert pl.plan(wp2(START), wp2(GOAL)) and pl.getResult().cost == P.traj_cost
    gpath = [w.pos[:2] for w in pl.getTraj().getWaypoints()]
    assert np.array_equal(np.array(gpath), np.array(path)[:, :2])

    def aux_of(mu):
        out = np.empty(int(np.prod(mu._dim)), dtype=np.int8)
        mu.ctx.check(mu.ctx.lib.mplx_aux_get(mu.ctx.h, out.ctypes.data))
        return out

    # ---- the perturbed plan: search region around the path + potential (distance_map_planner_node.cpp:175-193)
    Q = oracle(0.5)
    Q.set_search_region(path, (0.5, 0.5, 0))
    Q.set_potential_weights(10, 0)
    Q.update_potential_map((1.5, 1.5, 0), START)
    mu2, p2 = gpu(0.5)
    p2.setSearchRadius([0.5, 0.5]); p2.setSearchRegion(gpath)
    p2.setPotentialRadius([1.5, 1.5]); p2.setPotentialWeight(10); p2.setGradientWeight(0)
    p2.updatePotentialMap(START[:2])
    assert np.array_equal(aux_of(mu2), Q.aux_map().ravel())
    # get_succ: costs with the potential term, blocked flags at the region's border
    rng = np.random.default_rng(5)
    inside = np.argwhere(Q.aux_map()[0] >= 0)
    nodes = []
    for y, x in inside[rng.choice(len(inside), 60, replace=False)]:
        pos = ((x + 0.5) * Q._origin_res[1] + Q._origin_res[0][0], (y + 0.5) * Q._origin_res[1] + Q._origin_res[0][1], 0.0)
        nodes.append((pos, tuple(np.round(rng.uniform(-1.5, 1.5, 2), 1)) + (0.0,)))
    out = p2.getSuccBatch([util.gpu_wp(p, v) for p, v in nodes])
    nU = 9
    n_pot = 0
    for k, (p, v) in enumerate(nodes):
        succ, cost, act = Q.get_succ(orc.waypoint(p, v))
        got = [out[k * nU + i] for i in range(nU) if out[k * nU + i].valid]
        assert [g.action for g in got] == list(act)
        for g, co in zip(got, cost):
            assert g.cost == co or (np.isinf(g.cost) and np.isinf(co))
            n_pot += (not np.isinf(co)) and co > 11.0
    assert n_pot > 20  # the potential term really is in play
    r, c = util.compare_plan(Q, p2, (START, (0, 0, 0)), (GOAL,), orc.ACC)
    assert r.status == 0 and r.cost > P.traj_cost
    assert p2.kernelName() == "astar_spec_kernel<32,16,ACC,pot>"  # the POT build of the speculative kernel (9-input ACC lattice)
    p2.plan(util.gpu_wp(START, vel=(0, 0, 0)), util.gpu_wp(GOAL))  # (a second launch: the first one pays the code load)
    ms_spec = p2.lastKernelMs()
    # repeatability of the POT kernel (DESIGN: memory-ordering contract): the same plan five times, every result word equal
    word = lambda x: (x.status, x.cost, x.n_expanded, x.expand_hash, x.n_nodes, x.n_edges, x.voxel_reads, x.n_succ, x.n_succ_finite, x.n_push)
    again = set()
    for _ in range(5):
        p2.plan(util.gpu_wp(START, vel=(0, 0, 0)), util.gpu_wp(GOAL))
        again.add(word(p2.getResult()))
    assert again == {word(r)}
    p2.setSpeculation(0)  # ... and the one-node kernel: the same search, whole state space
    r1, _ = util.compare_plan(Q, p2, (START, (0, 0, 0)), (GOAL,), orc.ACC)
    assert p2.kernelName().startswith("astar_kernel<") and (r1.cost, r1.n_expanded, r1.expand_hash) == (r.cost, r.n_expanded, r.expand_hash)
    p2.plan(util.gpu_wp(START, vel=(0, 0, 0)), util.gpu_wp(GOAL))
    print(f"potential-field plan, {r.n_expanded} expansions: speculative kernel {ms_spec:.3f} ms, one-node kernel {p2.lastKernelMs():.3f} ms")
    p2.setSpeculation(-1)
    # getSearchRegion / getPotentialCloud
    a = Q.aux_map()[0]
    reg = p2.getSearchRegion()
    assert len(reg) == int((a >= 0).sum())
    cloud = p2.getPotentialCloud(1.0)
    assert len(cloud) == int(((a > 0) & (a < 100)).sum())
    cells = [Q.float_to_int((x, y, 0.0)) for x, y, _ in cloud[::50]]
    assert all(cloud[::50][i][2] == a[c[1], c[0]] / 100.0 for i, c in enumerate(cells))
    # ---- the potential alone on the whole map (distance_map_planner_node.cpp:201-224)
    R = oracle(0.5)
    R.set_potential_weights(10, 0)
    R.update_potential_map((1.5, 1.5, 0), START)
    mu3, p3 = gpu(0.5)
    p3.setPotentialRadius([1.5, 1.5]); p3.setPotentialWeight(10); p3.setGradientWeight(0)
    p3.updatePotentialMap(START[:2])
    assert np.array_equal(aux_of(mu3), R.aux_map().ravel())
    r3, _ = util.compare_plan(R, p3, (START, (0, 0, 0)), (GOAL,), orc.ACC)
    assert r3.status == 0 and P.traj_cost < r3.cost < r.cost
    # ---- a second planner on the same MapUtil without cost terms plans the plain search (nothing leaks)
    from mpl_ros_amd.planner import OccMapPlanner
    p4 = OccMapPlanner(False)
    p4.setMapUtil(mu3)
    p4.setVmax(2.0); p4.setAmax(1.0); p4.setDt(1.0); p4.setU(mapgen.control_lattice(1.0, 1, False)[:, :2]); p4.setTol(0.5)
    assert p4.plan(wp2(START), wp2(GOAL)) and p4.getResult().cost == oracle_cost(0.5)
    assert p3.plan(wp2(START), wp2(GOAL)) and p3.getResult().cost == r3.cost  # ... and p3 gets its potential back


@pytest.mark.gpu
def test_hip_potential_in_3d_with_a_range():
    """The same cost terms on a voxel map: spherical mask, potential restricted to a box around pos
    (setPotentialMapRange), 27-input lattice."""
    grid, origin, res = util.small_map(48, seed=21, occupancy=0.05)
    mapgen.carve_bubble(grid, (1.05, 1.05, 1.05), origin, res, 3)
    mapgen.carve_bubble(grid, (3.55, 3.55, 3.05), origin, res, 3)
    U = mapgen.control_lattice(1.0, 1, True)
    kw = dict(v_max=2.0, a_max=1.0, tol_pos=0.5)
    P = util.make_oracle(grid, origin, res, orc.ACC, U, **kw)
    P.set_potential_weights(3.0, 0)
    P.update_potential_map((0.4, 0.4, 0.3), (2.0, 2.0, 2.0), (1.5, 1.5, 1.0))
    mu, pl = util.make_gpu(grid, origin, res, U, **kw)
    pl.setPotentialRadius((0.4, 0.4, 0.3)); pl.setPotentialWeight(3.0); pl.setPotentialMapRange((1.5, 1.5, 1.0))
    pl.updatePotentialMap((2.0, 2.0, 2.0))
    out = np.empty(grid.size, dtype=np.int8)
    mu.ctx.check(mu.ctx.lib.mplx_aux_get(mu.ctx.h, out.ctypes.data))
    a = P.aux_map()
    assert np.array_equal(out, a.ravel()) and ((a > 0) & (a < 100)).sum() > 100
    r, c = util.compare_plan(P, pl, ((1.05, 1.05, 1.05), (0, 0, 0)), ((3.55, 3.55, 3.05),), orc.ACC)
    assert r.status == 0


@pytest.mark.gpu
@pytest.mark.parametrize("control,num,kernel", [(orc.JRK, 1, "astar_spec_kernel<32,16,JRK,pot>"), (orc.ACC, 2, "astar_spec_kernel<128,4,ACC,pot>"),
                                                (orc.JRK, 2, "astar_spec_kernel<128,4,JRK,pot>")])
def test_hip_potential_beyond_the_small_acc_lattices_runs_on_the_speculative_kernel(control, num, kernel):
    """Round 4: the POT builds of the speculative kernel for JRK states and for lattices up to 128 inputs (27 / 125 inputs
    here) -- the same cost terms, whole plans and state spaces against the oracle, and against the one-node kernel."""
    grid, origin, res = util.small_map(48, seed=21, occupancy=0.05)
    mapgen.carve_bubble(grid, (1.05, 1.05, 1.05), origin, res, 3)
    mapgen.carve_bubble(grid, (3.55, 3.55, 3.05), origin, res, 3)
    U = mapgen.control_lattice(1.0, num, True)
    kw = dict(v_max=2.0, a_max=1.0, tol_pos=0.5, max_expand=3000)
    if control == orc.JRK:
        kw["j_max"] = 1.0
    P = util.make_oracle(grid, origin, res, control, U, **kw)
    P.set_potential_weights(3.0, 0)
    P.update_potential_map((0.4, 0.4, 0.3), (2.0, 2.0, 2.0), (1.5, 1.5, 1.0))
    mu, pl = util.make_gpu(grid, origin, res, U, max_nodes=1 << 21, max_edges=1 << 23, **kw)
    pl.setPotentialRadius((0.4, 0.4, 0.3)); pl.setPotentialWeight(3.0); pl.setPotentialMapRange((1.5, 1.5, 1.0))
    pl.updatePotentialMap((2.0, 2.0, 2.0))
    start = ((1.05, 1.05, 1.05), (0, 0, 0), (0, 0, 0))
    r, c = util.compare_plan(P, pl, start, ((3.55, 3.55, 3.05),), control)
    assert pl.kernelName() == kernel and r.n_expanded > 20
    ms_spec = pl.lastKernelMs()
    pl.setSpeculation(0)
    r1, _ = util.compare_plan(P, pl, start, ((3.55, 3.55, 3.05),), control)
    assert pl.kernelName().startswith("astar_kernel<") and (r1.status, r1.cost, r1.n_expanded, r1.expand_hash, r1.n_nodes) == (r.status, r.cost, r.n_expanded, r.expand_hash, r.n_nodes)
    print(f"potential plan {kernel}: {r.n_expanded} expansions, {ms_spec:.2f} ms; one-node kernel {pl.lastKernelMs():.2f} ms")
