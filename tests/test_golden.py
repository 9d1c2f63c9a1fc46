"""Committed fixtures: the skir map (extracted from the reference's skir.bag), the `simple` map (BASELINE config 1 (ii):
re-rasterised from the reference's simple.stl by tests/golden/make_simple_fixture.py -- an approximate regeneration of
the lost simple.bag) and the plan results of tests/golden/plans.json / simple_plan.json.  CPU: the oracle reproduces
them; GPU: the HIP path reproduces them."""
import hashlib
import json
import os

import numpy as np
import pytest

from mpl_ros_amd import mapgen
from oracle import orc
from tests import util

HERE = os.path.dirname(os.path.abspath(__file__))
PLANS = json.load(open(os.path.join(HERE, "golden", "plans.json")))


def scenario_inputs(name, skir):
    if name == "skir_launch_query":
        grid, origin, res = skir
        return grid, origin, res, orc.ACC, 1, ((5.5, 5.5, 0.5), (1, 0, 0)), (1.5, 1.5, 5.5), dict(v_max=2.0, a_max=1.0, tol_pos=0.5)
    if name.startswith("acc96_seed"):
        seed = int(name[-1])
        grid, origin, res = util.small_map(96, seed=seed, occupancy=0.10)
        mapgen.carve_bubble(grid, (1.05, 1.05, 1.05), origin, res, 3)
        mapgen.carve_bubble(grid, (8.55, 8.55, 8.55), origin, res, 3)
        return grid, origin, res, orc.ACC, 1, ((1.05, 1.05, 1.05), (0, 0, 0)), (8.55, 8.55, 8.55), dict(v_max=2.0, a_max=1.0, tol_pos=0.5)
    grid, origin, res = util.small_map(96, seed=4, occupancy=0.10)
    mapgen.carve_bubble(grid, (1.05, 1.05, 1.05), origin, res, 3)
    return grid, origin, res, orc.JRK, 2, ((1.05, 1.05, 1.05), (0, 0, 0)), (8.55, 8.55, 8.55), dict(v_max=2.0, a_max=1.0, j_max=1.0, tol_pos=0.5, max_expand=3000)


def test_skir_fixture_is_the_reference_grid(skir):
    # sha256 of the int8 grid inside mpl_test_node/maps/skir/skir.bag
    grid, origin, res = skir
    assert grid.shape == (54, 99, 99)
    assert hashlib.sha256(grid.tobytes()).hexdigest() == "28e42b1a1c4d492a9f7517bb1fb3249991f6834f99cad63d3ef3bbaa250a3642"
    assert res == float(np.float32(0.1)) and origin[2] == float(np.float32(0.2))
    assert int((grid == 100).sum()) == 50409 and int((grid == 0).sum()) == 478845


def test_benchmark_map_generator_is_pinned():
    grid, origin, res, start, goal, rng = mapgen.benchmark_map(64)
    assert hashlib.sha256(grid.tobytes()).hexdigest() == "82ddb3378a01d381bfb3d9c5f66a7a450eaa6105058cc42399d985e804432c9c"
    assert abs((grid > 0).mean() - 0.10) < 0.01
    r = mapgen.SplitMix64(20250620)
    assert [r.next() for _ in range(2)] == [18070495504912055082, 16367535455934044849]
    assert mapgen.control_lattice(1, 1, True).shape == (27, 3) and mapgen.control_lattice(1, 2, True).shape == (125, 3)
    assert mapgen.control_lattice(1, 1, False).shape == (9, 3)


@pytest.mark.parametrize("plan", PLANS, ids=[p["name"] for p in PLANS])
def test_oracle_reproduces_golden_plans(plan, skir):
    grid, origin, res, control, num, start, goal, kw = scenario_inputs(plan["name"], skir)
    P = util.make_oracle(grid, origin, res, control, mapgen.control_lattice(1.0, num, True), **kw)
    st = P.plan(orc.waypoint(start[0], vel=start[1], control=control), orc.waypoint(goal, control=control))
    ids, _ = P.expanded()
    assert st == plan["status"] and len(ids) == plan["n_expanded"]
    assert str(util.expand_hash(ids)) == plan["expand_hash"]
    assert P.traj_cost == plan["cost"]
    tr = P.traj()
    assert tr["actions"].tolist() == plan["actions"] and tr["node_ids"].tolist() == plan["node_ids"]
    assert [orc.wp_state(w, control).tolist() for w in tr["wps"]] == plan["waypoints"]
    assert P.counters() == plan["counters"]


@pytest.mark.gpu
@pytest.mark.parametrize("plan", PLANS, ids=[p["name"] for p in PLANS])
def test_hip_reproduces_golden_plans(plan, skir):
    grid, origin, res, control, num, start, goal, kw = scenario_inputs(plan["name"], skir)
    mu, pl = util.make_gpu(grid, origin, res, mapgen.control_lattice(1.0, num, True), **kw)
    ok = pl.plan(util.gpu_wp(start[0], start[1], control=control), util.gpu_wp(goal, control=control))
    r = pl.getResult()
    assert ok and r.status == plan["status"] and r.n_expanded == plan["n_expanded"]
    assert str(r.expand_hash) == plan["expand_hash"] and r.cost == plan["cost"]
    tr = pl.getTraj()
    assert tr.actions.tolist() == plan["actions"] and tr.node_ids.tolist() == plan["node_ids"]
    assert [w.state().tolist() for w in tr.getWaypoints()] == plan["waypoints"]
    c = plan["counters"]
    assert (r.voxel_reads, r.n_succ, r.n_succ_finite, r.n_nodes) == (c["n_voxel_reads"], c["n_succ"], c["n_succ_finite"], c["n_new_nodes"])


# ---------------------------------------------------------------- BASELINE config 1 (ii): the `simple` map
SIMPLE_PLAN = json.load(open(os.path.join(HERE, "golden", "simple_plan.json")))
# launch/map_planner_node/test.launch:16-33, literally: start (14.5, 4.5, 0.05) at rest -> goal (2.4, 16.6, 0.05),
# v_max 2, a_max 1, yaw_max 0.5, u 1, u_yaw 0.5, dt 1, use_3d false (2-D lattice, nU = 9), use_yaw false; setTol(0.5)
SIMPLE_START, SIMPLE_GOAL = (14.5, 4.5, 0.05), (2.4, 16.6, 0.05)


@pytest.fixture(scope="module")
def simple_map():
    d = np.load(os.path.join(HERE, "golden", "simple_map.npz"))
    return d["grid"], tuple(d["origin"].tolist()), float(d["res"])


def test_simple_fixture_follows_the_mesh_to_map_recipe(simple_map):
    grid, origin, res = simple_map
    # cloud_to_map.cpp + mesh_to_map.launch.simple:21-27: res 0.1f, origin 0, range (18, 18, 2) -> int(18 / 0.1f) = 179, int(2 / 0.1f) = 19
    assert grid.shape == (19, 179, 179) and origin == (0.0, 0.0, 0.0) and res == float(np.float32(0.1))
    assert set(np.unique(grid).tolist()) == {0, 100}
    assert 0.03 < (grid > 0).mean() < 0.08
    assert grid[:, 0, :].all() and grid[:, :, 0].all()  # the boundary wall of the mesh


def test_oracle_reproduces_the_simple_launch_query(simple_map):
    grid, origin, res = simple_map
    U = mapgen.control_lattice(1.0, 1, False)
    assert U.shape == (9, 3)
    P = util.make_oracle(grid, origin, res, orc.ACC, U, v_max=2.0, a_max=1.0, tol_pos=0.5)
    st = P.plan(orc.waypoint(SIMPLE_START), orc.waypoint(SIMPLE_GOAL))
    ids, _ = P.expanded()
    plan = SIMPLE_PLAN
    assert st == plan["status"] == 0 and len(ids) == plan["n_expanded"] and str(util.expand_hash(ids)) == plan["expand_hash"]
    assert P.traj_cost == plan["cost"] and P.num_closed() == plan["n_closed"]
    tr = P.traj()
    assert tr["actions"].tolist() == plan["actions"] and tr["node_ids"].tolist() == plan["node_ids"]
    assert [orc.wp_state(w, orc.ACC).tolist() for w in tr["wps"]] == plan["waypoints"]
    assert P.counters() == plan["counters"]


@pytest.mark.gpu
def test_hip_plans_the_simple_launch_query_with_the_launch_file_parameters(simple_map):
    """Config 1 (ii) through the C-ABI with the launch file's own parameter set -- including setYawmax(0.5) with
    use_yaw = false, which must not change (or refuse) the plan -- against the oracle: the whole state space."""
    grid, origin, res = simple_map
    U = mapgen.control_lattice(1.0, 1, False)
    kw = dict(v_max=2.0, a_max=1.0, tol_pos=0.5)
    P = util.make_oracle(grid, origin, res, orc.ACC, U, **kw)
    mu, pl = util.make_gpu(grid, origin, res, U, **kw)
    pl.setYawmax(0.5)
    r, c = util.compare_plan(P, pl, (SIMPLE_START, (0, 0, 0)), (SIMPLE_GOAL,), orc.ACC)
    assert r.status == 0 and r.n_expanded == SIMPLE_PLAN["n_expanded"] and r.cost == SIMPLE_PLAN["cost"]
    assert str(r.expand_hash) == SIMPLE_PLAN["expand_hash"]
