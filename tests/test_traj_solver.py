"""The steps after the search: TrajSolver3D refinement and TrajectoryExtractor sampling.  Host
arithmetic: everything here runs without a GPU.  The product is the C++ of include/mpl_shim/mpl_traj_solver (through
the C-ABI: mplx_traj_solve / mplx_traj_sample, and compiled directly into tests/cpp/traj_refine_driver.cpp); the checker
is oracle/traj_ref.py (an independent KKT formulation) and closed forms.  Parity with upstream is unpinned (its sources
are absent and the reference holds no TrajSolver output)."""
import json
import math
import os
import subprocess

import numpy as np
import pytest

from mpl_ros_amd.planner import Control, Primitive3D, Trajectory3D, Waypoint3D
from mpl_ros_amd.traj_solver import TrajectoryExtractor, TrajSolver3D
from oracle import traj_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORDER = {Control.VEL: (0, 1), Control.ACC: (1, 2), Control.JRK: (2, 3)}


def wp(pos, vel=(0, 0, 0), acc=(0, 0, 0), control=Control.ACC):
    w = Waypoint3D(control)
    w.pos, w.vel, w.acc = np.array(pos, float), np.array(vel, float), np.array(acc, float)
    return w


def monomials(traj):
    return np.array([[traj_ref.primitive_to_monomials(p.coeff(k)) for k in range(3)] for p in traj.segs]).transpose(0, 2, 1)  # [seg][n][axis]


def fixed_of(ws, s):
    out = []
    for w in ws:
        f = {}
        for k, (bit, v) in enumerate(((1, w.pos), (2, w.vel), (4, w.acc), (8, w.jrk))):
            if k <= s and w.control & bit:
                f[k] = v
        out.append(f)
    return out


def test_rest_to_rest_single_segment_closed_forms():
    """One segment, both ends at rest: min velocity = the straight line, min acceleration = 3 tau^2 - 2 tau^3,
    min jerk = 10 tau^3 - 15 tau^4 + 6 tau^5 (tau = t / T)."""
    p0, p1, T = np.array([1.0, -2.0, 0.5]), np.array([4.0, 2.0, 0.0]), 2.5
    shapes = {Control.VEL: [0, 1, 0, 0, 0, 0], Control.ACC: [0, 0, 3, -2, 0, 0], Control.JRK: [0, 0, 0, 10, -15, 6]}
    for control, shape in shapes.items():
        ts = TrajSolver3D(control)
        ts.setWaypoints([wp(p0, control=control), wp(p1, control=control)])
        ts.setDts([T])
        tr = ts.solve()
        assert len(tr.segs) == 1 and tr.getTotalTime() == T
        a = monomials(tr)[0]
        want = np.array([[(p0[d] if n == 0 else 0.0) + (p1[d] - p0[d]) * shape[n] / T ** n for d in range(3)] for n in range(6)])
        assert np.allclose(a, want, rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize("control", [Control.VEL, Control.ACC, Control.JRK])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_refinement_is_the_constrained_minimiser(control, seed):
    """What map_planner_node.cpp:217-227 does with a search result: end states keep the search's control kind (ACC:
    position and velocity fixed), intermediate waypoints are set to Control::VEL (position only), segment times are the
    primitives' -- against the KKT solution of the same problem."""
    rng = np.random.default_rng(100 + seed)
    S = int(rng.integers(2, 9))
    ws = [wp(rng.uniform(-5, 5, 3), rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3), control=Control.ACC if i in (0, S) else Control.VEL) for i in range(S + 1)]
    if seed == 2:
        ws[0].control = Control.JRK  # a start with its acceleration fixed too
    dts = rng.uniform(0.5, 2.0, S).tolist()
    s, r = ORDER[control]
    ts = TrajSolver3D(control)
    ts.setWaypoints(ws)
    ts.setDts(dts)
    tr = ts.solve()
    assert len(tr.segs) == S and np.allclose(tr.getSegmentTimes(), dts)
    a = monomials(tr)
    assert np.all(a[:, 2 * (s + 1):, :] == 0)  # degree N - 1
    ref = traj_ref.solve([w.pos for w in ws], fixed_of(ws, s), dts, s, r)
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.allclose(a[:, :2 * (s + 1), :], ref, rtol=1e-7, atol=1e-8 * scale)
    j_prod = traj_ref.cost(a[:, :2 * (s + 1), :], dts, r)
    assert abs(j_prod - traj_ref.cost(ref, dts, r)) <= 1e-8 * max(1.0, j_prod)
    assert abs(tr.J(control) - j_prod) <= 1e-9 * max(1.0, j_prod)  # Trajectory::J is that integral
    # interpolation, fixed derivatives, continuity up to order s at the joints
    joints = tr.getWaypoints()
    for i, w in enumerate(ws):
        assert np.allclose(joints[i].pos, w.pos, atol=1e-9)
        if w.control & 2 and s >= 1:
            assert np.allclose(joints[i].vel, w.vel, atol=1e-9)
    for i in range(1, S):
        T = dts[i - 1]
        for k in range(s + 1):
            left = sum(traj_ref._falling(n, k) * a[i - 1, n] * T ** (n - k) for n in range(k, 6))
            assert np.allclose(left, math.factorial(k) * a[i, k], atol=1e-8 * scale)


def test_traj_solver_node_paths():
    """traj_solver_node.cpp:31-76: setPath + solve for VEL / ACC / JRK; the minimum-snap call "does not work" upstream
    either and yields no trajectory here."""
    path = [(0, 0, 0), (1, 0, 0), (2, 1, 0), (5, 1, 0)]
    costs = {}
    for control in (Control.VEL, Control.ACC, Control.JRK):
        ts = TrajSolver3D(control)
        ts.setPath(path)
        assert ts.getDts() == [1.0, 1.0, 3.0]  # L-infinity distances at v = 1
        tr = ts.solve()
        assert len(tr.segs) == 3 and tr.getTotalTime() == 5.0
        ws = tr.getWaypoints()
        assert all(np.allclose(w.pos, p, atol=1e-9) for w, p in zip(ws, path))
        if control != Control.VEL:
            assert np.allclose(ws[0].vel, 0, atol=1e-9) and np.allclose(ws[-1].vel, 0, atol=1e-9)
        costs[control] = tr.J(control)
    assert all(v > 0 for v in costs.values())
    ts = TrajSolver3D(Control.SNP)
    ts.setPath(path)
    assert ts.solve().segs == []
    ts = TrajSolver3D(Control.JRK)
    ts.setWaypoints([wp((0, 0, 0)), wp((1, 0, 0))])
    ts.setDts([1.0, 2.0])  # wrong count
    assert ts.solve().segs == []
    ts.setDts([0.0])  # a zero segment time
    assert ts.solve().segs == []


def test_trajectory_extractor_samples_like_the_reference_header():
    """trajectory_extractor.hpp:8-30: N = ceil(total / dt), N + 1 commands at i * total / N, carrying position ..
    jerk, yaw and yaw_dot."""
    prs = [Primitive3D([[0, 0, 0, 1.0, 0.0, 0.0], [0, 0, 0, 0.0, 1.0, 2.0], [0, 0, 0, 0, 0, 0.5]], 1.0, Control.ACC | 16, [0, 0, 0, 0, 0.5, 0.1]),
           Primitive3D([[0, 0, 0, -1.0, 1.0, 0.5], [0, 0, 0, 0.0, 1.0, 3.0], [0, 0, 0, 0, 0, 0.5]], 1.5, Control.ACC | 16, [0, 0, 0, 0, -0.5, 0.6])]
    tr = Trajectory3D(prs)
    cmds = TrajectoryExtractor(tr, 0.01).getCommands()
    N = math.ceil(2.5 / 0.01)
    assert len(cmds) == N + 1
    for i in (0, 1, 57, 100, 101, N):
        t = i * (2.5 / N)
        c = cmds[i]
        assert c.stamp == t
        seg, tau = (0, t) if t < 1.0 else (1, t - 1.0)
        cx = prs[seg].coeff(0)
        assert abs(c.position[0] - (cx[3] / 2 * tau * tau + cx[4] * tau + cx[5])) < 1e-12
        assert abs(c.velocity[0] - (cx[3] * tau + cx[4])) < 1e-12 and c.acceleration[0] == cx[3] and c.jerk[0] == 0
        cy = prs[seg].pr_yaw()
        assert abs(c.yaw - (cy[4] * tau + cy[5])) < 1e-12 and c.yaw_dot == cy[4]
    assert tr.Jyaw() == 0.25 * 1.0 + 0.25 * 1.5 and tr.J(Control.ACC) == 1.0 * 1.0 + 1.0 * 1.5


def test_cpp_refinement_through_the_drop_in_headers(tmp_path):
    """tests/cpp/traj_refine_driver.cpp = map_planner_node.cpp:205-227 after the plan (getWaypoints, intermediate
    control = VEL, getSegmentTimes, TrajSolver3D(Control::JRK), solve) on a fixed raw trajectory, compiled against the
    drop-in headers; where /root/reference exists the reference's OWN glue -- planning_ros_utils/primitive_ros_utils.h
    and trajectory_extractor.hpp, unchanged -- is compiled in as well and must round-trip and sample the result."""
    exe = str(tmp_path / "traj_refine_driver")
    ref = "/root/reference/planning_ros_utils"
    cmd = ["g++", "-std=c++14", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include", "mpl_shim"), "-I" + os.path.join(ROOT, "include")]
    with_ref = os.path.isdir(ref)
    if with_ref:
        cmd += ["-DMPLX_WITH_REFERENCE_GLUE", "-I" + ref + "/include", "-I" + ref + "/src/planning_utils"]
    out = subprocess.run(cmd + [os.path.join(ROOT, "tests", "cpp", "traj_refine_driver.cpp"), "-o", exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    # the same raw trajectory through the Python wrapper (C-ABI -> the same C++ classes): identical numbers
    u = [(1, 0, 0), (1, 1, 0), (0, 1, 1), (-1, 0, 0), (0, -1, -1)]
    w = wp((1, 2, 0.5))
    prs = []
    for ui in u:
        prs.append(Primitive3D([[0, 0, 0, ui[k], w.vel[k], w.pos[k]] for k in range(3)], 1.0, Control.ACC))
        w = wp(w.pos + w.vel + np.array(ui) / 2.0, w.vel + np.array(ui))
    raw = Trajectory3D(prs)
    ws = raw.getWaypoints()
    for x in ws:
        x.control = Control.ACC
    for x in ws[1:-1]:
        x.control = Control.VEL
    ts = TrajSolver3D(Control.JRK)
    ts.setWaypoints(ws)
    ts.setDts(raw.getSegmentTimes())
    tr = ts.solve()
    assert r["raw_J"] == [raw.J(c) for c in (Control.VEL, Control.ACC, Control.JRK, Control.SNP)]
    assert r["refined_J"] == [tr.J(c) for c in (Control.VEL, Control.ACC, Control.JRK, Control.SNP)]
    assert r["coeff"] == [[list(p.coeff(k)) for k in range(3)] for p in tr.segs]
    assert r["refined_J"][2] < r["raw_J"][2] or r["raw_J"][2] == 0  # jerk effort went down (the raw ACC primitives have none inside, but jump at the joints)
    assert r["with_reference_glue"] == with_ref
    if with_ref:
        cmds = TrajectoryExtractor(tr, 0.01).getCommands()
        assert r["n_cmds"] == len(cmds) and r["roundtrip_equal"]
        assert r["cmd_mid"] == [cmds[len(cmds) // 2].stamp] + list(cmds[len(cmds) // 2].position) + list(cmds[len(cmds) // 2].velocity)
