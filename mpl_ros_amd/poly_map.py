"""Host mirror of the reference's moving-obstacle (PolyMap) planner interface over the C-ABI (mplx_poly_*).

Names follow mpl_external_planner/include/mpl_external_planner/poly_map_planner/ (poly_map_planner.h:18-60,
simple_obstacle.h): a `PolyWorld` is what ONE PolyMapPlanner2D sees -- setMap(ori, dim), setStartTime,
setStaticObstacles / setLinearObstacles / setNonlinearObstacles -- and a `PolyTeam` holds the worlds of all robots of
a decentralised tick (robot_team.hpp:33-66) so that their expansions run in one launch.  No compute happens here.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import MplxError

VEL, ACC, JRK, SNP = _capi.VEL, _capi.ACC, _capi.JRK, _capi.SNP


def rectangle(hx, hy=None):
    """Polyhedron2D of an axis-aligned box with half sizes (hx, hy) around the origin, hyperplanes in the order of
    multi_robot_node.cpp:65-69: rows {px, py, nx, ny}."""
    hy = hx if hy is None else hy
    return np.array([[-hx, 0, -1, -0.0], [hx, 0, 1, 0], [0, -hy, -0.0, -1], [0, hy, 0, 1]], dtype=np.float64)


class StaticObstacle:          # PolyhedronObstacle2D(poly, p)
    def __init__(self, poly, p):
        self.poly, self.p = np.ascontiguousarray(poly, dtype=np.float64).reshape(-1, 4), np.array(p, dtype=np.float64)


class LinearObstacle:          # PolyhedronLinearObstacle2D(poly, p, v) + set_cov_v
    def __init__(self, poly, p, v, cov_v=0.0):
        self.poly, self.p, self.v, self.cov_v = np.ascontiguousarray(poly, dtype=np.float64).reshape(-1, 4), np.array(p, float), np.array(v, float), float(cov_v)


class NonlinearObstacle:       # PolyhedronNonlinearObstacle2D(poly, traj, t) + disappear_front_/back_
    def __init__(self, poly, segs, start_t, disappear_front=False, disappear_back=False):
        """segs: rows {cx[6], cy[6], T} -- the primitives of the obstacle's trajectory"""
        self.poly = np.ascontiguousarray(poly, dtype=np.float64).reshape(-1, 4)
        self.segs = np.ascontiguousarray(segs, dtype=np.float64).reshape(-1, 13)
        self.start_t, self.disappear_front, self.disappear_back = float(start_t), bool(disappear_front), bool(disappear_back)


class PolyWorld:
    def __init__(self, ori, dim, start_t=0.0):
        self.ori, self.dim, self.start_t = np.array(ori, float), np.array(dim, float), float(start_t)
        self.static, self.linear, self.nonlinear = [], [], []


U9 = np.array([(dx, dy) for dx in (-1.0, 0.0, 1.0) for dy in (-1.0, 0.0, 1.0)])  # multi_robot_node.cpp:56-59 with u = 1, num = 1

# Team2 (robot_team.hpp:275-353): start / goal of the 16 robots on the border of the 10 m x 10 m map
TEAM2 = [((0, -5), (10, 5)), ((0, -2.5), (10, 2.5)), ((0, 0), (10, 0)), ((0, 2.5), (10, -2.5)), ((0, 5), (10, -5)), ((2.5, 5), (7.5, -5)),
         ((5, 5), (5, -5)), ((7.5, 5), (2.5, -5)), ((10, 5), (0, -5)), ((10, 2.5), (0, -2.5)), ((10, 0), (0, 0)), ((10, -2.5), (0, 2.5)),
         ((10, -5), (0, 5)), ((7.5, -5), (2.5, 5)), ((5, -5), (5, 5)), ((2.5, -5), (7.5, 5))]


def acc_segs(p0, v0, us, dt):
    """Trajectory of ACC primitives from (p0, v0) under the inputs `us`: rows {cx[6], cy[6], T}."""
    p, v = np.array(p0, float), np.array(v0, float)
    rows = []
    for u in us:
        u = np.array(u, float)
        rows.append([0, 0, 0, u[0], v[0], p[0], 0, 0, 0, u[1], v[1], p[1], dt])
        p = u / 2 * dt * dt + v * dt + p
        v = u * dt + v
    return np.array(rows)


def jrk_segs(p0, v0, a0, us, dt):
    """Trajectory of JRK primitives from (p0, v0, a0) under the jerk inputs `us` (what a JRK robot's plan looks like to
    the others: cubic segments, poly_map_planner_node.cpp:73-85 use_acc): rows {cx[6], cy[6], T}."""
    p, v, a = np.array(p0, float), np.array(v0, float), np.array(a0, float)
    rows = []
    for u in us:
        u = np.array(u, float)
        rows.append([0, 0, u[0], a[0], v[0], p[0], 0, 0, u[1], a[1], v[1], p[1], dt])
        p = u / 6 * dt ** 3 + a / 2 * dt * dt + v * dt + p
        v = u / 2 * dt * dt + a * dt + v
        a = u * dt + a
    return np.array(rows)


def team2_tick(dt=0.5, t_now=1.0, traj_time=4.0):
    """One decentralised replanning tick of Team2 (BASELINE config 5): HomogeneousRobotTeam::set_obs
    (robot_team.hpp:33-51) gives robot i the static box (robot_team.hpp:383-388) and the 15 other robots as
    nonlinear obstacles following their current trajectories, truncated to `traj_time` (robot.hpp:156-170); every
    robot then plans from traj.evaluate(dt) to its goal (robot.hpp:92-133).  The robots' current trajectories are
    synthetic stand-ins (accelerate towards the goal, coast, brake).  Returns (worlds, starts, goals)."""
    rec = rectangle(0.5)
    box = np.array([[4, 0, -1, -0.0], [6, 0, 1, 0], [5, -1, -0.0, -1], [5, 1, 0, 1]], dtype=np.float64)
    trajs, traj_t = [], []
    for s, g in TEAM2:
        d = np.sign(np.array(g, float) - np.array(s, float))
        trajs.append(acc_segs(s, (0, 0), [d, d, 0 * d, 0 * d, 0 * d, 0 * d, -d, -d], dt))
        traj_t.append(0.01 * len(traj_t))
    worlds = []
    for i in range(16):
        W = PolyWorld((0.0, -5.0), (10.0, 10.0))
        W.static.append(StaticObstacle(box, (0.0, 0.0)))
        for j in range(16):
            if j == i:
                continue
            segs, dis = trajs[j], False
            if traj_time > 0 and len(segs) * dt > traj_time:
                segs, dis = segs[: int(round(traj_time / dt))], True
            W.nonlinear.append(NonlinearObstacle(rec, segs, start_t=t_now - traj_t[j], disappear_back=dis))
        worlds.append(W)
    starts, goals = np.zeros((16, 9)), np.zeros((16, 9))
    for r, (s, g) in enumerate(TEAM2):
        seg = trajs[r][1]
        starts[r] = [seg[5], seg[11], seg[4], seg[10], 0, 0, 0, 0, dt]
        goals[r, 0:2] = g
    return worlds, starts, goals


REPLANNER_OBS = [((6, 12), (0, -0.6)), ((10, 6), (0, 0.5)), ((13, 14), (-0.3, -0.7)), ((16, 9), (0, 0.4)), ((8, 9.5), (0.4, 0.0))]


def replanner_world(t, turn=False, scale=1):
    """Synthetic world of the replanner flow (poly_map_replanner_node.cpp: setLinearObstacles(obstacles as they are at t) +
    setStartTime(t) per replan), seen at time t: a (20 scale) m square map with 2 m boxes moving at constant velocity
    (PolyhedronLinearObstacle2D, cov_v 0.2) -- five of them at scale 1; at scale 2 the same five at twice the coordinates plus
    five more, shifted.  turn: from t = 2 on the last box of a set moves the other way and the second one stops -- primitives
    the planner had found free get blocked, blocked ones get free.  Start / goal: replanner_endpoints(scale)."""
    W = PolyWorld((0.0, 0.0), (20.0 * scale, 20.0 * scale), start_t=t)
    rec = rectangle(1.0)
    sets = [(float(scale), (0.0, 0.0))] + ([(float(scale), (7.0, -5.0))] if scale > 1 else [])
    for sc, off in sets:
        for k, (p, v) in enumerate(REPLANNER_OBS):
            p, v = np.array(p, float) * sc + np.array(off), np.array(v, float)
            pos = p + v * t
            if turn and t >= 2.0:
                if k == 4:
                    pos = p + v * 2.0 - v * (t - 2.0)
                    v = -v
                if k == 1:
                    pos = p + v * 2.0
                    v = 0 * v
            W.linear.append(LinearObstacle(rec, pos, v, cov_v=0.2))
    return W


def replanner_endpoints(scale=1):
    start = np.zeros(9); start[0], start[1] = 0.5, 10.0 * scale
    goal = np.zeros(9); goal[0], goal[1] = 20.0 * scale - 1.0, 10.0 * scale
    return start, goal


class PolyTeam:
    """The worlds of several planners on the device + the shared planner set-up (setVmax/setAmax/setDt/setU/setW)."""

    def __init__(self, device=0):
        self.lib = _capi.load()
        self.h = C.c_void_p()
        code = self.lib.mplx_poly_create(device, C.byref(self.h))
        if code != _capi.OK:
            raise MplxError(self.lib.mplx_poly_last_error(None).decode())
        self.n_u = 0

    def __del__(self):
        try:
            if self.h:
                self.lib.mplx_poly_destroy(self.h)
        except Exception:
            pass

    def check(self, code):
        if code != _capi.OK:
            raise MplxError(self.lib.mplx_poly_last_error(self.h).decode())

    def configure(self, control, U, dt, v_max=-1.0, a_max=-1.0, j_max=-1.0, w=10.0):
        U = np.ascontiguousarray(U, dtype=np.float64).reshape(-1, 2)
        self.n_u = U.shape[0]
        self.check(self.lib.mplx_poly_config(self.h, int(control), self.n_u, U.ctypes.data, float(dt), float(v_max), float(a_max), float(j_max), float(w)))

    def set_worlds(self, worlds):
        self.check(self.lib.mplx_poly_begin(self.h, len(worlds)))
        D2 = C.c_double * 2
        for i, W in enumerate(worlds):
            self.check(self.lib.mplx_poly_set_world(self.h, i, D2(*W.ori), D2(*W.dim), W.start_t))
            for o in W.static:
                self.check(self.lib.mplx_poly_add_static(self.h, i, len(o.poly), o.poly.ctypes.data, D2(*o.p)))
            for o in W.linear:
                self.check(self.lib.mplx_poly_add_linear(self.h, i, len(o.poly), o.poly.ctypes.data, D2(*o.p), D2(*o.v), o.cov_v))
            for o in W.nonlinear:
                self.check(self.lib.mplx_poly_add_nonlinear(self.h, i, len(o.poly), o.poly.ctypes.data, len(o.segs), o.segs.ctypes.data,
                                                            o.start_t, int(o.disappear_front), int(o.disappear_back)))
        self.check(self.lib.mplx_poly_commit(self.h))

    def get_succ_batch(self, world_of, states):
        """env_poly_map::get_succ for K nodes: states K x 9 (pos2 vel2 acc2 jrk2 t); returns K x n_u records."""
        w = np.ascontiguousarray(world_of, dtype=np.int32)
        s = np.ascontiguousarray(states, dtype=np.float64).reshape(-1, 9)
        out = (_capi.PolySucc * (len(w) * self.n_u))()
        self.check(self.lib.mplx_poly_get_succ_batch(self.h, len(w), w.ctypes.data, s.ctypes.data, out))
        return out

    def set_capacity(self, n_slots, nodes, edges, open_log):
        self.check(self.lib.mplx_poly_set_capacity(self.h, int(n_slots), int(nodes), int(edges), int(open_log)))

    def set_helpers(self, per_leader):
        """Look-ahead helper workgroups per leader (-1 auto, 0 off): results are identical with or without."""
        self.check(self.lib.mplx_poly_set_helpers(self.h, int(per_leader)))

    def set_deadline(self, seconds):
        """Launch guard: a tick that outlives `seconds` is aborted and plan_batch raises MplxError (MPLX_ERR_TIMEOUT)."""
        self.check(self.lib.mplx_poly_set_deadline(self.h, float(seconds)))

    def last_helpers(self):
        return int(self.lib.mplx_poly_last_helpers(self.h))

    def set_record(self, cap):
        self.check(self.lib.mplx_poly_set_record(self.h, int(cap)))

    def plan_batch(self, world_of, starts, goals, eps=1.0, tol_pos=0.5, tol_vel=-1.0, max_expand=-1, heur_ignore_dynamics=True):
        """PlannerBase::plan for one query per entry, all in one launch; starts / goals: n x 9 (pos2 vel2 acc2 jrk2 t)."""
        w = np.ascontiguousarray(world_of, dtype=np.int32)
        s = np.ascontiguousarray(starts, dtype=np.float64).reshape(-1, 9)
        g = np.ascontiguousarray(goals, dtype=np.float64).reshape(-1, 9)
        R = (_capi.Result * len(w))()
        self.check(self.lib.mplx_poly_plan_batch(self.h, len(w), w.ctypes.data, s.ctypes.data, g.ctypes.data, float(eps), float(tol_pos), float(tol_vel),
                                                 int(max_expand), int(bool(heur_ignore_dynamics)), R))
        self._results = [R[i] for i in range(len(w))]
        return self._results

    def traj(self, q):
        r = self._results[q]
        n = r.traj_len if r.status == _capi.PLAN_OK else 0
        act = np.zeros(max(n, 1), dtype=np.int32); ids = np.zeros(n + 1, dtype=np.int32); st = np.zeros((n + 1, 9))
        if n:
            self.check(self.lib.mplx_poly_result_traj(self.h, q, act.ctypes.data, ids.ctypes.data, st.ctypes.data))
        return act[:n], ids[:n + 1] if n else ids[:0], st[:n + 1] if n else st[:0]

    def expanded_ids(self, q):
        cap = int(self._results[q].n_expanded)
        ids = np.zeros(max(cap, 1), dtype=np.int32)
        n = C.c_uint32()
        self.check(self.lib.mplx_poly_result_expanded(self.h, q, cap, ids.ctypes.data, C.byref(n)))
        return ids[:n.value]

    def cycles(self, q):
        """shader-clock cycles query q spent in pop / get_succ / look-up + commit"""
        cyc = (C.c_uint64 * 10)()
        self.check(self.lib.mplx_poly_result_cycles(self.h, int(q), cyc))
        return dict(pop=int(cyc[0]), get_succ=int(cyc[1]), commit=int(cyc[2]), primitives=int(cyc[3]), start_test=int(cyc[4]), prepare=int(cyc[5]), items_lane0=int(cyc[6]), items_wait=int(cyc[7]), lookahead_hits=int(cyc[8]))

    def lpa(self):
        """A PolyLpa on this team's worlds: the LPA* planner of poly_map_replanner_node.cpp (setLPAstar(true), updateNodes, getSubStateSpace)."""
        return PolyLpa(self)

    def last_kernel_ms(self):
        ms = C.c_float()
        self.check(self.lib.mplx_poly_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value


# ---------------------------------------------------------------------------------------------
# The reference's multi-robot loop (mpl_test_node/src/robot.hpp:92-170, robot_team.hpp:33-66,366-388, multi_robot_node.cpp:
# 95-105), host logic over any planner with a plan(world, start, goal) -> (status, actions, states) callable: the robots
# plan once without obstacles at t = 0, ddt, 2 ddt, ..., then every loop tick set_obs(time) gives robot i the static
# box and the 15 other robots' CURRENT trajectories as nonlinear obstacles, and a robot replans once dt has passed since
# its last plan (with ddt = 0.01 one robot per 0.01 s tick; with ddt = 0 all sixteen in the same tick: the batched tick).
class Robot:
    def __init__(self, start, goal, dt):
        self.start = np.array([start[0], start[1], 0.0, 0.0, 0, 0, 0, 0, 0.0])  # pos2 vel2 acc2 jrk2 t (t is set per plan)
        self.goal = np.array([goal[0], goal[1], 0.0, 0.0, 0, 0, 0, 0, 0.0])
        self.dt = dt
        self.segs = np.zeros((0, 13))   # the primitives of the current trajectory: rows {cx[6], cy[6], T}
        self.traj_t = -10000.0          # robot.hpp:218

    def state_after_first_primitive(self):
        """traj_.evaluate(dt_) (robot.hpp:97): Trajectory::evaluate picks the segment with taus[id] <= tau < taus[id + 1]
        -- for tau = dt that is the second primitive at local time 0 -- or the last one at its end."""
        n = len(self.segs)
        if n == 0:
            return None
        taus = np.concatenate([[0.0], np.cumsum(self.segs[:, 12])])
        tau = min(max(self.dt, 0.0), taus[-1])
        for i in range(n):
            if (taus[i] <= tau < taus[i + 1]) or i + 1 == n:
                lt = tau - taus[i]
                cx, cy = self.segs[i, 0:6], self.segs[i, 6:12]
                p = [c[0] / 120 * lt * lt * lt * lt * lt + c[1] / 24 * lt * lt * lt * lt + c[2] / 6 * lt * lt * lt + c[3] / 2 * lt * lt + c[4] * lt + c[5] for c in (cx, cy)]
                v = [c[0] / 24 * lt * lt * lt * lt + c[1] / 6 * lt * lt * lt + c[2] / 2 * lt * lt + c[3] * lt + c[4] for c in (cx, cy)]
                return p, v
        return None

    def obstacle(self, t, max_t, shape):
        """get_nonlinear_obstacle(t, max_t) (robot.hpp:156-170)"""
        segs, dis = self.segs, False
        if max_t > 0 and float(np.sum(segs[:, 12])) > max_t:
            segs, dis = segs[: int(round(max_t / self.dt))], True
        return NonlinearObstacle(shape, segs, start_t=t - self.traj_t, disappear_back=dis)


class RobotTeam:
    """HomogeneousRobotTeam<2> / Team2 (robot_team.hpp).  plan_many(worlds, starts, goals) -> list of (status, actions,
    states (n + 1) x 9) plans the robots that are due in this tick -- with the device planner: in one launch."""

    def __init__(self, layout=TEAM2, ddt=0.01, dt=0.5, traj_time=0.0, origin=(0.0, -5.0), dim=(10.0, 10.0)):
        self.robots = [Robot(s, g, dt) for s, g in layout]
        self.ddt, self.dt, self.traj_time, self.origin, self.dim = ddt, dt, traj_time, origin, dim
        self.shape = rectangle(0.5)
        self.static = []  # (the static box is added AFTER the robots' first plans: robot_team.hpp:366-388)
        self.box = np.array([[4, 0, -1, -0.0], [6, 0, 1, 0], [5, -1, -0.0, -1], [5, 1, 0, 1]], dtype=np.float64)
        self.plans = 0

    def _world_for(self, i, time, with_obs):
        W = PolyWorld(self.origin, self.dim)
        if with_obs:
            W.static = list(self.static)
            for j, r in enumerate(self.robots):
                if j != i:
                    W.nonlinear.append(r.obstacle(time, self.traj_time, self.shape))
        return W

    def _adopt(self, r, time, res, U):
        status, actions, states = res
        if status != 0:
            return False
        rows = []
        for a, s in zip(actions, states[:-1]):
            u = U[int(a)]
            rows.append([0, 0, 0, u[0], s[2], s[0], 0, 0, 0, u[1], s[3], s[1], self.dt])
        r.segs = np.array(rows).reshape(-1, 13)
        r.traj_t = time
        self.plans += 1
        return True

    def init(self, plan_many, U):
        """Team2::init (robot_team.hpp:366-376): robot i plans at t = i ddt, no obstacles set yet"""
        t = 0.0
        for i, r in enumerate(self.robots):
            s = r.start.copy()
            s[8] = 0.0
            if not self._adopt(r, t, plan_many([self._world_for(i, t, False)], [s], [r.goal])[0], U):
                return False
            t += self.ddt
        self.static = [StaticObstacle(self.box, (0.0, 0.0))]
        return True

    def update_decentralized(self, time, plan_many, U):
        """set_obs(time), then every robot's plan(time) (robot_team.hpp:60-66, robot.hpp:92-133); the obstacle sets are
        taken before anybody replans, so the robots due in this tick are independent and go to the planner together"""
        due, worlds, starts, goals = [], [], [], []
        for i, r in enumerate(self.robots):
            pv = r.state_after_first_primitive()
            if r.traj_t >= 0 and pv is not None:
                r.start[0:2], r.start[2:4] = pv[0], pv[1]
            if time - r.traj_t < self.dt - 1e-8 or float(np.linalg.norm(r.start[0:2] - r.goal[0:2])) < 1:
                continue
            s = r.start.copy()
            s[8] = self.dt  # (start_ = traj_.evaluate(dt_) carries t = dt_: Trajectory::evaluate stamps the query time)
            due.append(i); worlds.append(self._world_for(i, time, True)); starts.append(s); goals.append(r.goal)
        if not due:
            return True, []
        res = plan_many(worlds, starts, goals)
        ok = all(self._adopt(self.robots[i], time, rr, U) for i, rr in zip(due, res))
        return ok, due


class PolyLpa:
    """PolyMapPlanner2D with setLPAstar(true) (poly_map_replanner_node.cpp:341-352): a device-resident LPA* state space over the
    moving-obstacle environment of ONE world of a PolyTeam (mplx_plpa_*).  The flow of replanCallback / plan() there:
    team.set_worlds(...) (setLinearObstacles, setStartTime) -> update_nodes() -> plan(start, goal) -> sub_state_space(1)."""

    def __init__(self, team, world=0):
        self.team, self.lib, self.world = team, team.lib, int(world)
        h = C.c_void_p()
        code = self.lib.mplx_plpa_create(team.h, C.byref(h))
        if code != _capi.OK:
            raise MplxError(f"mplx_plpa_create failed ({code})")
        self.h = h
        self.result = None

    def __del__(self):
        try:
            if self.h:
                self.lib.mplx_plpa_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def check(self, code):
        if code != _capi.OK:
            raise MplxError(f"mplx error {code}: {self.lib.mplx_plpa_last_error(self.h).decode()}")

    def set_capacity(self, nodes, edges, open_log):
        self.check(self.lib.mplx_plpa_set_capacity(self.h, int(nodes), int(edges), int(open_log)))

    def initialized(self):
        return bool(self.lib.mplx_plpa_initialized(self.h))

    def reset(self):
        self.check(self.lib.mplx_plpa_reset(self.h))

    def plan(self, start, goal, eps=1.0, tol_pos=0.5, tol_vel=-1.0, max_expand=-1, heur_ignore_dynamics=True):
        s = np.ascontiguousarray(start, dtype=np.float64); g = np.ascontiguousarray(goal, dtype=np.float64)
        R = _capi.Result()
        self.check(self.lib.mplx_plpa_plan(self.h, self.world, s.ctypes.data, g.ctypes.data, float(eps), float(tol_pos), float(tol_vel), int(max_expand),
                                           int(bool(heur_ignore_dynamics)), C.byref(R)))
        self.result = R
        return R.status == _capi.PLAN_OK

    def update_nodes(self):
        """PolyMapPlanner::updateNodes -> (entries that became blocked, entries that became free, [(entry, now blocked)] by entry number)"""
        nb, nc, n = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.check(self.lib.mplx_plpa_update_nodes(self.h, self.world, C.byref(nb), C.byref(nc)))
        self.check(self.lib.mplx_plpa_changed(self.h, 0, None, None, C.byref(n)))
        e = np.zeros(max(n.value, 1), dtype=np.int32); b = np.zeros(max(n.value, 1), dtype=np.int32)
        self.check(self.lib.mplx_plpa_changed(self.h, n.value, e.ctypes.data, b.ctypes.data, C.byref(n)))
        return int(nb.value), int(nc.value), list(zip(e[:n.value].tolist(), b[:n.value].tolist()))

    def sub_state_space(self, time_step):
        self.check(self.lib.mplx_plpa_sub_state_space(self.h, self.world, int(time_step)))

    def traj(self):
        n = int(self.lib.mplx_plpa_traj_len(self.h))
        act = np.zeros(max(n, 1), dtype=np.int32); ids = np.zeros(n + 1, dtype=np.int32); st = np.zeros((n + 1, 9))
        if n:
            self.check(self.lib.mplx_plpa_result_traj(self.h, act.ctypes.data, ids.ctypes.data, st.ctypes.data))
        return act[:n], ids[:n + 1] if n else ids[:0], st[:n + 1] if n else st[:0]

    def expanded_ids(self):
        cap = int(self.result.n_expanded) if self.result is not None else 0
        ids = np.zeros(max(cap, 1), dtype=np.int32)
        n = C.c_uint32()
        self.check(self.lib.mplx_plpa_result_expanded(self.h, cap, ids.ctypes.data, C.byref(n)))
        return ids[:n.value]

    def last_kernel_ms(self):
        ms = C.c_float()
        self.check(self.lib.mplx_plpa_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value

    def cycles(self):
        """shader cycles of the last plan by section of an iteration (thread 0's clock)"""
        cyc = (C.c_uint64 * 10)()
        self.check(self.lib.mplx_plpa_result_cycles(self.h, cyc))
        names = ("pop", "stop_test_settle", "primitives_lookups_heuristics", "is_free", "link", "update_children", "goal_test_barrier")
        return {n: int(cyc[i]) for i, n in enumerate(names)}

    def state_space(self):
        nn, ne = C.c_uint64(), C.c_uint64()
        self.check(self.lib.mplx_plpa_counts(self.h, C.byref(nn), C.byref(ne)))
        n, m = int(nn.value), int(ne.value)
        states = np.zeros((max(n, 1), 9)); g = np.zeros(max(n, 1)); rhs = np.zeros(max(n, 1)); h = np.zeros(max(n, 1))
        closed = np.zeros(max(n, 1), dtype=np.int32); opened = closed.copy(); built = closed.copy()
        self.check(self.lib.mplx_plpa_result_nodes(self.h, max(n, 1), states.ctypes.data, g.ctypes.data, rhs.ctypes.data, h.ctypes.data, closed.ctypes.data,
                                                   opened.ctypes.data, built.ctypes.data))
        child = np.zeros(max(m, 1), dtype=np.int32); parent = child.copy(); action = child.copy(); blocked = child.copy()
        self.check(self.lib.mplx_plpa_result_entries(self.h, max(m, 1), child.ctypes.data, parent.ctypes.data, action.ctypes.data, blocked.ctypes.data))
        return dict(n_nodes=n, states=states[:n], g=g[:n], rhs=rhs[:n], h=h[:n], closed=closed[:n], opened=opened[:n], built=built[:n],
                    child=child[:m], parent=parent[:m], action=action[:m], blocked=blocked[:m], initialized=self.initialized())
