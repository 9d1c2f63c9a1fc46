"""After the search: TrajSolver3D refinement (map_planner_node.cpp:217-227, traj_solver_node.cpp:
40-76) and TrajectoryExtractor sampling (planning_ros_utils/src/planning_utils/trajectory_extractor.hpp:8-30).

Host arithmetic behind the C-ABI (mplx_traj_solve / mplx_traj_sample): the same C++ classes a C++ caller gets from the
drop-in headers (include/mpl_shim/mpl_traj_solver/traj_solver.h, mpl_basis/trajectory.h)."""
import ctypes as C
import math

import numpy as np

from . import _capi
from ._capi import MplxError
from .planner import Control, Primitive3D, Trajectory3D, Waypoint3D


class TrajSolver3D:
    def __init__(self, control, debug=False):
        self._control = int(control)
        self._wps, self._dts, self._path, self._v = [], [], [], 1.0

    def setWaypoints(self, ws):
        self._wps = list(ws)

    def setDts(self, dts):
        self._dts = [float(t) for t in dts]

    def setV(self, v):
        self._v = float(v)

    def setPath(self, path):
        """Positions only: both ends at rest in every derivative of the control kind, the points between fixed in position
        only, segment times = L-infinity distance / v."""
        self._path = [np.asarray(p, dtype=np.float64) for p in path]
        self._wps, self._dts = [], []
        for i, p in enumerate(self._path):
            w = Waypoint3D(self._control & 15 if i in (0, len(self._path) - 1) else Control.VEL)
            w.pos = p.copy()
            self._wps.append(w)
            if i:
                self._dts.append(float(np.max(np.abs(p - self._path[i - 1])) / self._v))

    def getWaypoints(self):
        return self._wps

    def getDts(self):
        return self._dts

    def solve(self, verbose=False):
        n = len(self._wps)
        if n < 2 or len(self._dts) != n - 1:
            if verbose:
                print(f"[TrajSolver] {n} waypoints need {max(n - 1, 0)} segment times, got {len(self._dts)}")
            return Trajectory3D([])
        W = (_capi.Waypoint * n)(*[w.to_c() for w in self._wps])
        D = (C.c_double * (n - 1))(*self._dts)
        P = (_capi.Primitive * (n - 1))()
        if _capi.load().mplx_traj_solve(self._control, n, W, D, P) != _capi.OK:
            print("\x1b[31m[TrajSolver] no trajectory: a segment time <= 0, a singular system, or a minimum-snap request "
                  "(septic segments do not fit a Primitive)\x1b[0m")
            return Trajectory3D([])
        return Trajectory3D([Primitive3D.from_c(P[i]) for i in range(n - 1)])


class TrajectoryCommand:
    """planning_ros_msgs/TrajectoryCommand: stamp (seconds from the first sample), position, velocity, acceleration,
    jerk, yaw, yaw_dot."""
    __slots__ = ("stamp", "position", "velocity", "acceleration", "jerk", "yaw", "yaw_dot")


class TrajectoryExtractor:
    """TrajectoryExtractor(traj, dt): N = ceil(total time / dt) -> N + 1 commands (trajectory_extractor.hpp:8-30)."""

    def __init__(self, traj, dt):
        N = int(math.ceil(traj.getTotalTime() / dt))
        self._cmds = []
        for w in traj.sample(N):
            c = TrajectoryCommand()
            c.stamp, c.position, c.velocity, c.acceleration, c.jerk = w.t, w.pos, w.vel, w.acc, w.jrk
            c.yaw, c.yaw_dot = w.yaw, w.yaw_dot
            self._cmds.append(c)

    def getCommands(self):
        return self._cmds
