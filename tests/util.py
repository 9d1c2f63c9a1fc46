"""Shared helpers of the parity tests: build the oracle and the HIP planner on the same inputs."""
import numpy as np

from mpl_ros_amd import mapgen
from oracle import orc


def small_map(n=64, seed=7, occupancy=0.08):
    grid, _ = mapgen.random_box_map((n, n, n), seed=seed, occupancy=occupancy, edge=(2, 8))
    return grid, (0.0, 0.0, 0.0), 0.1


def make_oracle(grid, origin, res, control, U, **kw):
    P = orc.Planner()
    P.set_map(grid, origin, res)
    P.free_unknown()
    P.set_config(control, U, **kw)
    return P


def make_gpu(grid, origin, res, U, v_max=-1.0, a_max=-1.0, j_max=-1.0, dt=1.0, w=10.0, eps=1.0, tol_pos=0.5,
             tol_vel=-1.0, tol_acc=-1.0, max_expand=-1, heur_ignore_dynamics=False, t_max=float("inf"),
             n_slots=1, max_nodes=1 << 20, max_edges=1 << 22, max_log=1 << 21, record=0, spec=-1, yaw_max=-1.0):
    from mpl_ros_amd.planner import VoxelMapPlanner, VoxelMapUtil
    mu = VoxelMapUtil()
    dz, dy, dx = grid.shape
    mu.setMap(origin, (dx, dy, dz), grid.ravel(), res)
    mu.freeUnknown()
    pl = VoxelMapPlanner(False)
    pl.setMapUtil(mu)
    pl.setVmax(v_max); pl.setAmax(a_max); pl.setJmax(j_max); pl.setDt(dt); pl.setW(w); pl.setEpsilon(eps)
    pl.setTol(tol_pos, tol_vel, tol_acc); pl.setMaxNum(max_expand); pl.setHeurIgnoreDynamics(heur_ignore_dynamics)
    pl.setTmax(t_max)
    pl.setYawmax(yaw_max)
    pl.setU(U)
    pl.setCapacity(n_slots, max_nodes, max_edges, max_log)
    pl.setSpeculation(spec)
    if record:
        pl.setRecord(record)
    return mu, pl


def gpu_wp(pos, vel=(0, 0, 0), acc=(0, 0, 0), jrk=(0, 0, 0), control=orc.ACC, t=0.0, yaw=None):
    from mpl_ros_amd.planner import Waypoint3D
    w = Waypoint3D(control)
    if yaw is not None:
        w.use_yaw, w.yaw = True, float(yaw)
    w.pos, w.vel, w.acc, w.jrk = np.array(pos, float), np.array(vel, float), np.array(acc, float), np.array(jrk, float)
    w.t = t
    return w


def expand_hash(ids):
    h = 0
    for i in ids:
        h = (h * 0x100000001B3 + (int(i) + 1)) & ((1 << 64) - 1)
    return h


def random_states(rng, n, control, lo, hi, v_max=2.0, a_max=1.0):
    """n random states (pos uniform in [lo,hi]^3, vel/acc on the 0.1 lattice like search states)."""
    out = []
    for _ in range(n):
        pos = np.round(rng.uniform(lo, hi, 3), 2)
        vel = np.round(rng.uniform(-v_max, v_max, 3), 1)
        acc = np.round(rng.uniform(-a_max, a_max, 3), 1) if control & 4 else np.zeros(3)
        jrk = np.round(rng.uniform(-1, 1, 3), 1) if control & 8 else np.zeros(3)
        out.append((pos, vel, acc, jrk))
    return out


def compare_plan(P, pl, start, goal, control, check_traj=True, yaw=None):
    """Run the same query on the oracle (P) and on the HIP planner (pl); assert bit-exact agreement.
    yaw: (start yaw, goal yaw) makes the states yaw-carrying (use_yaw)."""
    ys, yg = yaw if yaw is not None else (None, None)
    so = orc.waypoint(start[0], vel=start[1], acc=start[2] if len(start) > 2 else (0, 0, 0), control=control, yaw=ys)
    go = orc.waypoint(goal[0], vel=goal[1] if len(goal) > 1 else (0, 0, 0), control=control, yaw=yg)
    P.reset_counters()
    st_o = P.plan(so, go)
    sg = gpu_wp(start[0], vel=start[1], acc=start[2] if len(start) > 2 else (0, 0, 0), control=control, yaw=ys)
    gg = gpu_wp(goal[0], vel=goal[1] if len(goal) > 1 else (0, 0, 0), control=control, yaw=yg)
    ok = pl.plan(sg, gg)
    r = pl.getResult()
    assert r.status == st_o, (r.status, st_o)
    assert ok == (st_o == orc.OK)
    c = P.counters()
    ids_o, _ = P.expanded()
    assert r.n_expanded == c["n_expansions"] == len(ids_o)
    assert r.expand_hash == expand_hash(ids_o)
    assert r.n_nodes == P.num_nodes()
    assert r.n_closed == P.num_closed()
    assert r.voxel_reads == c["n_voxel_reads"]
    assert r.n_succ == c["n_succ"] and r.n_succ_finite == c["n_succ_finite"]
    assert r.n_primitives == c["n_primitives"]
    assert r.n_reopen == c["n_reopen"]
    # the whole state space: predecessor lists (every node, edges in arrival order), g, h, closed flags
    co, po, ao = P.edges()
    cg, pg, ag = pl.getEdges()
    assert r.n_edges == len(co) == len(cg)
    assert np.array_equal(cg, co) and np.array_equal(pg, po) and np.array_equal(ag, ao)
    if r.n_nodes <= 60000:
        _, _, g_g, h_g, closed_g, _ = pl._nodes()
        g_o = np.empty(r.n_nodes); h_o = np.empty(r.n_nodes); closed_o = np.empty(r.n_nodes, dtype=np.int32)
        for i in range(r.n_nodes):
            _, g_o[i], h_o[i], closed_o[i] = P.node(i)
        assert np.array_equal(g_g, g_o) and np.array_equal(h_g, h_o) and np.array_equal(closed_g, closed_o)
    if st_o == orc.OK:
        assert r.cost == P.traj_cost  # bit-exact f64
    else:
        assert np.isinf(r.cost)
    if check_traj and st_o == orc.OK:
        to = P.traj()
        tg = pl.getTraj()
        assert len(tg.segs) == to["n"]
        assert np.array_equal(tg.actions, to["actions"])
        assert np.array_equal(tg.node_ids, to["node_ids"])
        for wg, wo in zip(tg.getWaypoints(), to["wps"]):
            assert wg.control == wo.control
            assert np.array_equal(wg.state(), orc.wp_state(wo, wo.control))  # bit-exact (<= 1e-6 required); yaw included when carried
        for pg, po in zip(tg.segs, to["prs"]):
            for k in range(3):
                assert np.array_equal(pg.coeff(k), np.array(po.c[k][:]))
            assert np.array_equal(pg.pr_yaw(), np.array(po.cyaw[:]))
    return r, c
