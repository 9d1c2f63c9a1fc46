"""Host-side mirror of the reference's planner interface for the voxel-map path, over the C-ABI.

Same names, argument meaning and error behaviour as the C++ classes the reference drivers use
(mpl_test_node/src/map_planner_node.cpp:10-35,155-196):

    map_util = VoxelMapUtil(); map_util.setMap(origin, dim, data, res); map_util.freeUnknown()
    planner = VoxelMapPlanner(verbose); planner.setMapUtil(map_util); planner.setVmax(..) ...
    ok = planner.plan(start, goal); traj = planner.getTraj(); planner.getCloseSet()

Every call that computes goes through libmplx.so (HIP, gfx950).  Nothing here evaluates
primitives, touches voxels or searches on the CPU.
"""
import ctypes as C
import sys
import weakref
import math

import numpy as np

from . import _capi
from ._capi import ACC, JRK, SNP, VEL, MplxError  # noqa: F401  (re-exported Control kinds)


class Control:
    VEL, ACC, JRK, SNP = VEL, ACC, JRK, SNP


class Waypoint3D:
    """Waypoint<3>: pos/vel/acc/jrk + use_* flags (control = union of the flags)."""

    def __init__(self, control=0):
        self.pos = np.zeros(3)
        self.vel = np.zeros(3)
        self.acc = np.zeros(3)
        self.jrk = np.zeros(3)
        self.yaw = 0.0
        self.t = 0.0
        self.enable_t = False
        self.control = control

    # use_pos .. use_jrk view the control bits, like the union in the reference's Waypoint
    def _bit(self, b):
        return bool(self.control & b)

    def _set(self, b, v):
        self.control = (self.control | b) if v else (self.control & ~b)

    use_pos = property(lambda s: s._bit(1), lambda s, v: s._set(1, v))
    use_vel = property(lambda s: s._bit(2), lambda s, v: s._set(2, v))
    use_acc = property(lambda s: s._bit(4), lambda s, v: s._set(4, v))
    use_jrk = property(lambda s: s._bit(8), lambda s, v: s._set(8, v))
    use_yaw = property(lambda s: s._bit(16), lambda s, v: s._set(16, v))  # (map_planner_node.cpp:165)

    def to_c(self):
        w = _capi.Waypoint()
        w.pos[:] = [float(x) for x in self.pos]
        w.vel[:] = [float(x) for x in self.vel]
        w.acc[:] = [float(x) for x in self.acc]
        w.jrk[:] = [float(x) for x in self.jrk]
        w.yaw, w.t = float(self.yaw), float(self.t)
        w.control = int(self.control)
        w.enable_t = int(bool(self.enable_t))
        return w

    @staticmethod
    def from_c(w):
        o = Waypoint3D(w.control)
        o.pos, o.vel = np.array(w.pos[:]), np.array(w.vel[:])
        o.acc, o.jrk = np.array(w.acc[:]), np.array(w.jrk[:])
        o.yaw, o.t, o.enable_t = w.yaw, w.t, bool(w.enable_t)
        return o

    def state(self):
        parts = [self.pos]
        if self.control & 2:
            parts.append(self.vel)
        if self.control & 4:
            parts.append(self.acc)
        if self.control & 8:
            parts.append(self.jrk)
        if self.control & 16:
            parts.append(np.array([self.yaw]))
        return np.concatenate(parts)


class Primitive3D:
    """Primitive<3> as its coefficient table (pr(i).coeff(), t(), control())."""

    def __init__(self, coeffs, t, control, yaw_coeff=None):
        self._c = np.array(coeffs, dtype=np.float64).reshape(3, 6)
        self._t = float(t)
        self._control = control
        self._cyaw = np.zeros(6) if yaw_coeff is None else np.array(yaw_coeff, dtype=np.float64)

    def pr_yaw(self):
        """Coefficients of the yaw channel (use_yaw primitives): yaw(t) = c[4] t + c[5]."""
        return self._cyaw

    def coeff(self, i):
        return self._c[i]

    def t(self):
        return self._t

    def control(self):
        return self._control

    def to_c(self):
        p = _capi.Primitive()
        for k in range(3):
            p.c[k][:] = [float(x) for x in self._c[k]]
        p.cyaw[:] = [float(x) for x in self._cyaw]
        p.t, p.control = self._t, int(self._control)
        return p

    @staticmethod
    def from_c(p):
        return Primitive3D([list(p.c[k]) for k in range(3)], p.t, p.control, list(p.cyaw))


class Trajectory3D:
    def __init__(self, prs, wps=None, actions=None, cost=math.nan):
        self.segs = prs
        self._wps = wps
        self.actions = actions
        self.cost = cost

    def _c_prs(self):
        return (_capi.Primitive * max(len(self.segs), 1))(*[p.to_c() for p in self.segs])

    def sample(self, N):
        """Trajectory::sample(N) (trajectory_extractor.hpp:9-10): N + 1 equally spaced states; each carries .yaw_dot."""
        if not self.segs or N <= 0:
            return []
        out = (_capi.Waypoint * (N + 1))()
        yd = np.zeros(N + 1)
        _check(None, _capi.load().mplx_traj_sample(len(self.segs), self._c_prs(), int(N), out, yd.ctypes.data), _capi.load())
        ws = [Waypoint3D.from_c(out[i]) for i in range(N + 1)]
        for w, y in zip(ws, yd):
            w.yaw_dot = float(y)
        return ws

    def J(self, control):
        """Trajectory::J(control): total effort of the derivative `control` selects (map_planner_node.cpp:210-214)."""
        return float(_capi.load().mplx_traj_effort(len(self.segs), self._c_prs(), int(control) & 15))

    def Jyaw(self):
        return float(_capi.load().mplx_traj_effort(len(self.segs), self._c_prs(), 16))

    def getPrimitives(self):
        return self.segs

    def getWaypoints(self):
        """States at the segment joints (map_planner_node.cpp:217); a search result carries the search's own states."""
        if self._wps is None:
            ts = np.concatenate([[0.0], np.cumsum([p.t() for p in self.segs])])
            ws = []
            for i, p in enumerate(self.segs + self.segs[-1:]):
                one = Trajectory3D([p]).sample(1)[0 if i < len(self.segs) else 1]
                one.t = float(ts[i])
                ws.append(one)
            self._wps = ws
        return self._wps

    def getSegmentTimes(self):
        return [p.t() for p in self.segs]

    def getTotalTime(self):
        return float(sum(p.t() for p in self.segs))


def _check(ctx, code, lib):
    if code != _capi.OK:
        msg = lib.mplx_last_error(ctx)
        raise MplxError(f"mplx error {code}: {msg.decode() if msg else ''}")


class _Context:
    """One device context (HIP stream + map replica + pools)."""

    def __init__(self, device=0):
        self.lib = _capi.load()
        h = C.c_void_p()
        code = self.lib.mplx_ctx_create(device, C.byref(h))
        if code != _capi.OK:
            msg = self.lib.mplx_last_error(None)
            raise MplxError(f"mplx_ctx_create failed ({code}): {msg.decode() if msg else ''}")
        self.h = h

    def check(self, code):
        _check(self.h, code, self.lib)

    def __del__(self):
        try:
            if self.h:
                self.lib.mplx_ctx_destroy(self.h)
                self.h = None
        except Exception:
            pass


class _Lpa:
    """The device-resident state space of one LPA* planner (mplx_lpa).  Holds its context alive: the handle must be
    destroyed before the context."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.lib = ctx.lib
        h = C.c_void_p()
        code = self.lib.mplx_lpa_create(ctx.h, C.byref(h))
        if code != _capi.OK:
            raise MplxError(f"mplx_lpa_create failed ({code})")
        self.h = h

    def check(self, code):
        if code != _capi.OK:
            msg = self.lib.mplx_lpa_last_error(self.h)
            raise MplxError(f"mplx error {code}: {msg.decode() if msg else ''}")

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.lib.mplx_lpa_destroy(self.h)
            self.h = None
        except Exception:
            pass


class VoxelMapUtil:
    """MapUtil<3>: the voxel grid lives in HBM; setMap copies it in (or adopts a device pointer)."""

    def __init__(self, device=0):
        self.ctx = _Context(device)
        self._dim = None

    def setMap(self, ori, dim, data, res):
        """data: int8 sequence, x fastest (idx = x + dx*y + dx*dy*z); free 0, occupied >0, unknown -1."""
        a = np.ascontiguousarray(data, dtype=np.int8).ravel()
        if a.size != int(dim[0]) * int(dim[1]) * int(dim[2]):
            raise ValueError("map size does not match dim")
        d = (C.c_int32 * 3)(*[int(x) for x in dim])
        o = (C.c_double * 3)(*[float(x) for x in ori])
        self.ctx.check(self.ctx.lib.mplx_map_set(self.ctx.h, a.ctypes.data, d, o, float(res)))
        self._dim = tuple(int(x) for x in dim)

    def setMapDevice(self, device_ptr, ori, dim, res):
        """Adopt a grid already in HBM (e.g. a torch uint8/int8 CUDA tensor's data_ptr())."""
        d = (C.c_int32 * 3)(*[int(x) for x in dim])
        o = (C.c_double * 3)(*[float(x) for x in ori])
        self.ctx.check(self.ctx.lib.mplx_map_set_device(self.ctx.h, C.c_void_p(device_ptr), d, o, float(res)))
        self._dim = tuple(int(x) for x in dim)

    def _info(self):
        d = (C.c_int32 * 3)()
        o = (C.c_double * 3)()
        r = C.c_double()
        self.ctx.check(self.ctx.lib.mplx_map_info(self.ctx.h, d, o, C.byref(r)))
        return np.array(d[:]), np.array(o[:]), r.value

    def getDim(self):
        return self._info()[0]

    def getOrigin(self):
        return self._info()[1]

    def getRes(self):
        return self._info()[2]

    def getMap(self):
        n = int(np.prod(self._dim))
        out = np.empty(n, dtype=np.int8)
        self.ctx.check(self.ctx.lib.mplx_map_get(self.ctx.h, out.ctypes.data))
        return out

    def freeUnknown(self):
        self.ctx.check(self.ctx.lib.mplx_map_free_unknown(self.ctx.h))

    def query(self, pts):
        """Batched floatToInt + cell state for points (n,3): returns (cells int32 (n,3), state int8)
        with state 0 free, 1 occupied, 2 unknown, 3 outside."""
        p = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 3)
        cells = np.empty((p.shape[0], 3), dtype=np.int32)
        st = np.empty(p.shape[0], dtype=np.int8)
        self.ctx.check(self.ctx.lib.mplx_map_query(self.ctx.h, p.shape[0], p.ctypes.data, cells.ctypes.data, st.ctypes.data))
        return cells, st

    def floatToInt(self, pt):
        return self.query([pt])[0][0]

    def isFree(self, pt):
        return int(self.query([pt])[1][0]) == 0

    def isOccupied(self, pt):
        return int(self.query([pt])[1][0]) == 1

    def isOutside(self, pt):
        return int(self.query([pt])[1][0]) == 3

    def isUnknown(self, pt):
        return int(self.query([pt])[1][0]) == 2

    # ---- integer-cell forms (MapUtil::isFree(const Veci&) ..., map_replanner_node.cpp:180,217)
    def cellStates(self, cells):
        """state (0 free, 1 occupied, 2 unknown, 3 outside) of n cells (n,3) in one launch"""
        c = np.ascontiguousarray(cells, dtype=np.int32).reshape(-1, 3)
        st = np.empty(c.shape[0], dtype=np.int8)
        self.ctx.check(self.ctx.lib.mplx_map_cells(self.ctx.h, c.shape[0], c.ctypes.data, st.ctypes.data))
        return st

    def isFreeCell(self, pn):
        return int(self.cellStates([pn])[0]) == 0

    def isOccupiedCell(self, pn):
        return int(self.cellStates([pn])[0]) == 1

    def dilate(self, neighbours):
        """MapUtil::dilate(const vec_Veci&), map_planner_node.cpp:75-85"""
        o = np.ascontiguousarray(neighbours, dtype=np.int32).reshape(-1, 3)
        self.ctx.check(self.ctx.lib.mplx_map_dilate(self.ctx.h, o.shape[0], o.ctypes.data))

    def rayTrace(self, pt1, pt2):
        """MapUtil::rayTrace, map_replanner_node.cpp:177,208: cells (n,3) between two points"""
        a = (C.c_double * 3)(*[float(v) for v in pt1]); b = (C.c_double * 3)(*[float(v) for v in pt2])
        n = C.c_int(0)
        self.ctx.check(self.ctx.lib.mplx_map_raytrace(self.ctx.h, a, b, None, 0, C.byref(n)))
        out = np.empty((max(n.value, 1), 3), dtype=np.int32)
        self.ctx.check(self.ctx.lib.mplx_map_raytrace(self.ctx.h, a, b, out.ctypes.data, n.value, C.byref(n)))
        return out[:n.value]

    def _cloud(self, which):
        n = C.c_uint64(0)
        self.ctx.check(self.ctx.lib.mplx_map_cloud(self.ctx.h, which, None, 0, C.byref(n)))
        out = np.empty((max(n.value, 1), 3), dtype=np.float64)
        self.ctx.check(self.ctx.lib.mplx_map_cloud(self.ctx.h, which, out.ctypes.data, n.value, C.byref(n)))
        return out[:n.value]

    def getCloud(self):          # map_display.cpp:244
        return self._cloud(0)

    def getFreeCloud(self):      # map_display.cpp:256
        return self._cloud(1)

    def getUnknownCloud(self):   # map_display.cpp:266
        return self._cloud(2)


class VoxelMapPlanner:
    """PlannerBase<3, Waypoint3D> + MapPlanner<3> over the device back-end."""

    def __init__(self, verbose=False):
        self.planner_verbose_ = verbose
        self.map_util_ = None
        self._U = None
        self._U_yaw = None
        self._v_max = self._a_max = self._j_max = self._yaw_max = -1.0
        self._dt = 1.0
        self._w = 10.0
        self._eps = 1.0
        self._tol = (0.5, -1.0, -1.0)
        self._t_max = math.inf
        self._max_num = -1
        self._heur_ignore_dynamics = False
        self._dirty = True
        self._control = None
        self._result = None
        self._results = None
        self.traj_cost_ = math.inf
        self._use_lpastar = False
        self._lpa = None
        self._cap = None
        self._record = 0

    # ---- setters (PlannerBase / MapPlanner)
    def setMapUtil(self, map_util):
        self.map_util_ = map_util
        self._dirty = True

    def setVmax(self, v):
        self._v_max, self._dirty = float(v), True

    def setAmax(self, a):
        self._a_max, self._dirty = float(a), True

    def setJmax(self, j):
        self._j_max, self._dirty = float(j), True

    def setDt(self, dt):
        self._dt, self._dirty = float(dt), True

    def setW(self, w):
        self._w, self._dirty = float(w), True

    def setEpsilon(self, eps):
        self._eps, self._dirty = float(eps), True

    def setMaxNum(self, n):
        self._max_num, self._dirty = int(n), True

    def setTmax(self, t):
        self._t_max, self._dirty = float(t), True

    def setHeurIgnoreDynamics(self, ignore):
        self._heur_ignore_dynamics, self._dirty = bool(ignore), True

    def setU(self, U):
        U = np.asarray(U, dtype=np.float64)
        self._U_yaw = None
        if U.ndim == 2 and U.shape[1] == 4:
            # Vec4f control inputs (x, y, z, yaw rate) of the use_yaw lattices, map_planner_node.cpp:119-139
            self._U_yaw = np.ascontiguousarray(U[:, 3], dtype=np.float64)
            U = U[:, :3]
        self._U = np.ascontiguousarray(U, dtype=np.float64).reshape(-1, 3)
        self._dirty = True

    def setYawmax(self, yaw_max):
        """setYawmax (map_planner_node.cpp:179-180).  The threshold constrains yaw-carrying primitives only: the
        reference node always calls it, and its config-1 launch file passes yaw_max = 0.5 with use_yaw = false
        (launch/map_planner_node/test.launch:28,33).  It takes effect when the search states carry yaw (use_yaw)."""
        self._yaw_max, self._dirty = float(yaw_max), True

    # ---- LPA* incremental replanning: the state space stays on the device between plan() calls
    def setLPAstar(self, use_lpastar):
        """setLPAstar (map_replanner_node.cpp:425,437): plan() repairs and re-uses the state space of the previous
        plan (mplx_lpa_*: ComputeShortestPath of LPA* on a device-resident state space of this planner's own)."""
        self._use_lpastar = bool(use_lpastar)
        if not use_lpastar:
            self._lpa = None

    def _lpa_handle(self):
        ctx = self._ctx()
        if self._lpa is None or self._lpa.ctx is not ctx:
            self._lpa = _Lpa(ctx)
            if self._cap is not None:
                self._lpa.check(ctx.lib.mplx_lpa_set_capacity(self._lpa.h, self._cap[1], self._cap[2], self._cap[3]))
            if getattr(self, "_reroot", None) is not None:
                self._lpa.check(ctx.lib.mplx_lpa_set_reroot(self._lpa.h, self._reroot))
        return self._lpa

    def setSubStateSpaceMode(self, mode):
        """How getSubStateSpace re-roots (mplx_lpa_set_reroot): 0 Dijkstra through the expanded states of the space being left,
        1 an A* from the new root kept as the new space, 2 (default) auto -- 1 for spaces of more than 16384 states."""
        self._reroot = int(mode)
        if self._lpa is not None:
            self._lpa.check(self._lpa.lib.mplx_lpa_set_reroot(self._lpa.h, self._reroot))

    def initialized(self):
        """PlannerBase::initialized() (map_replanner_node.cpp:195,232,244): an LPA* state space exists."""
        return bool(self._lpa is not None and self._lpa.lib.mplx_lpa_initialized(self._lpa.h))

    def reset(self):
        if self._lpa is not None:
            self._lpa.check(self._lpa.lib.mplx_lpa_reset(self._lpa.h))

    def _update_nodes(self, fn, pns):
        if not self._use_lpastar or self._lpa is None:
            return 0
        self._configure(self._control)  # (another planner may have configured the shared context since)
        c = np.ascontiguousarray(pns, dtype=np.int32).reshape(-1, 3)
        n = C.c_uint64(0)
        self._lpa.check(getattr(self._lpa.lib, fn)(self._lpa.h, c.shape[0], c.ctypes.data, C.byref(n)))
        return int(n.value)

    def updateBlockedNodes(self, blocked_pns):   # map_replanner_node.cpp:196
        """After the shared MapUtil was edited: predecessor entries whose primitive is no longer free get cost inf
        (increaseCost).  Returns the number of entries that changed (upstream returns their primitives)."""
        return self._update_nodes("mplx_lpa_update_blocked", blocked_pns)

    def updateClearedNodes(self, cleared_pns):   # map_replanner_node.cpp:233
        return self._update_nodes("mplx_lpa_update_cleared", cleared_pns)

    def getSubStateSpace(self, time_step):       # map_replanner_node.cpp:245
        """Re-root the LPA* state space at the time_step-th state of the last trajectory; the caller then plans from
        getTraj().getWaypoints()[time_step] (map_replanner_node.cpp:246-250)."""
        if not self._use_lpastar or self._lpa is None:
            return
        self._configure(self._control)
        self._lpa.check(self._lpa.lib.mplx_lpa_sub_state_space(self._lpa.h, int(time_step)))

    # ---- potential-field cost and search region (distance_map_planner_node.cpp:185-193,199,218-224,231)
    # The auxiliary map lives on the MapUtil's device context; a planner re-sends its own when another planner sharing
    # the MapUtil has replaced (or removed) it since.
    @staticmethod
    def _v3(v):
        v = [float(x) for x in v]
        return v + [0.0] * (3 - len(v))

    def setSearchRadius(self, radius):           # distance_map_planner_node.cpp:185
        self._search_radius = self._v3(radius)

    def setSearchRegion(self, path, dense=False):  # distance_map_planner_node.cpp:186
        """The voxels within the search radius of `path` (waypoint positions; joined up with rayTrace unless dense):
        primitives that leave the region are blocked."""
        pts = np.array([self._v3(p) for p in path], dtype=np.float64).reshape(-1, 3)
        self._region = (pts, bool(dense))
        self._aux_dirty = True

    def setPotentialRadius(self, radius):        # distance_map_planner_node.cpp:187
        self._pot_radius = self._v3(radius)

    def setPotentialMapRange(self, range_):      # (commented out at distance_map_planner_node.cpp:190,221)
        self._pot_range = self._v3(range_)

    def setPotentialWeight(self, w):             # distance_map_planner_node.cpp:188
        self._pot_weight = float(w)
        self._aux_dirty = True

    def setGradientWeight(self, w):              # distance_map_planner_node.cpp:189
        if float(w) != 0.0:
            raise MplxError("setGradientWeight: only 0 (the value the reference passes) is supported by this back-end")
        self._grad_weight = 0.0

    def updatePotentialMap(self, pos, range_=None):  # distance_map_planner_node.cpp:191
        """Build the potential map around the obstacles of the CURRENT map (setPotentialRadius first)."""
        if getattr(self, "_pot_radius", None) is None:
            raise MplxError("setPotentialRadius first")
        self._pot_call = (self._v3(pos), self._v3(range_) if range_ is not None else getattr(self, "_pot_range", [0.0, 0.0, 0.0]))
        self._aux_dirty = True
        self._apply_aux()

    _aux_ids = iter(range(1, 1 << 62))

    def _aux_id(self):
        """This planner's owner tag for the context's auxiliary map: never reused (not id(self)); bit 63 tells the Python
        wrapper's tags from the C++ shim's."""
        if getattr(self, "_aux_id_v", None) is None:
            self._aux_id_v = (1 << 63) | next(VoxelMapPlanner._aux_ids)
        return self._aux_id_v

    def _has_aux(self):
        return getattr(self, "_region", None) is not None or getattr(self, "_pot_call", None) is not None

    def _apply_aux(self):
        """Make the context's auxiliary map this planner's: region first, then the potential (which keeps the region)."""
        ctx = self._ctx()
        # the owner tag lives on the context itself (mplx_aux_token): the library resets it whenever it drops the auxiliary
        # map (another grid, mplx_potential_clear), so "still mine" can never outlive the map it refers to
        tok = C.c_uint64(0)
        ctx.check(ctx.lib.mplx_aux_token(ctx.h, 0, 0, C.byref(tok)))
        me = self._aux_id()
        mine = tok.value == me
        if not self._has_aux():
            if tok.value != 0 and not mine:
                ctx.check(ctx.lib.mplx_potential_clear(ctx.h))  # another planner's cost terms must not leak into this plan
            return
        if mine and not getattr(self, "_aux_dirty", True):
            return
        d3 = lambda v: (C.c_double * 3)(*v)
        if not mine:
            ctx.check(ctx.lib.mplx_potential_clear(ctx.h))
        region = getattr(self, "_region", None)
        if region is not None:
            pts, dense = region
            if getattr(self, "_search_radius", None) is None:
                raise MplxError("setSearchRadius first")
            ctx.check(ctx.lib.mplx_search_region_set(ctx.h, pts.shape[0], pts.ctypes.data, d3(self._search_radius), int(dense)))
        ctx.check(ctx.lib.mplx_potential_weights(ctx.h, getattr(self, "_pot_weight", 0.0), 0.0))
        call = getattr(self, "_pot_call", None)
        if call is not None:
            pos, rng = call
            ctx.check(ctx.lib.mplx_potential_update(ctx.h, d3(self._pot_radius), d3(pos), d3(rng), 1))
        ctx.check(ctx.lib.mplx_aux_token(ctx.h, me, 1, None))
        self._aux_dirty = False

    def _aux_cloud(self, which):
        ctx = self._ctx()
        n = C.c_uint64(0)
        ctx.check(ctx.lib.mplx_aux_cloud(ctx.h, which, None, None, 0, C.byref(n)))
        pts = np.empty((max(n.value, 1), 3), dtype=np.float64)
        vals = np.empty(max(n.value, 1), dtype=np.int8)
        ctx.check(ctx.lib.mplx_aux_cloud(ctx.h, which, pts.ctypes.data, vals.ctypes.data, n.value, C.byref(n)))
        return pts[:n.value], vals[:n.value]

    def getPotentialCloud(self, h_max=1.0):      # distance_map_planner_node.cpp:231
        """Voxels with a potential strictly between 0 and 100 as (x, y, z) points; the 2-D planner lifts them to
        z = h_max * potential / 100 (a height field for display)."""
        self._apply_aux()
        pts, vals = self._aux_cloud(0)
        if isinstance(self, OccMapPlanner):
            pts = pts.copy()
            pts[:, 2] = float(h_max) * vals.astype(np.float64) / 100.0
        return pts

    def getSearchRegion(self):                   # distance_map_planner_node.cpp:199
        self._apply_aux()
        return self._aux_cloud(1)[0]

    def setTol(self, tol_pos, tol_vel=-1.0, tol_acc=-1.0):
        self._tol, self._dirty = (float(tol_pos), float(tol_vel), float(tol_acc)), True

    # ---- device-side capacities (no reference counterpart: the reference grows std containers)
    def setCapacity(self, n_slots=0, max_nodes=0, max_edges=0, max_open_log=0):
        ctx = self._ctx()
        ctx.check(ctx.lib.mplx_set_capacity(ctx.h, n_slots, max_nodes, max_edges, max_open_log))
        self._cap = (n_slots, max_nodes, max_edges, max_open_log)
        if self._lpa is not None:
            self._lpa.check(ctx.lib.mplx_lpa_set_capacity(self._lpa.h, max_nodes, max_edges, max_open_log))

    def setBucketWidth(self, width):
        ctx = self._ctx()
        ctx.check(ctx.lib.mplx_set_bucket_width(ctx.h, float(width)))

    def setSpeculation(self, mode):
        """-1 auto, 0 sequential kernel, 2 speculative multi-node expansion (same results)."""
        ctx = self._ctx()
        ctx.check(ctx.lib.mplx_set_speculation(ctx.h, int(mode)))

    def setHelpers(self, per_leader=-1, reserved=-1, cache_rows=0):
        """Helper workgroups (look-ahead expansion on idle compute units).  per_leader: -1 auto (4; 2 for the 65..128-input jerk lattices), 0 off, 2..4 (several
        helpers of one leader split its wish list by record index); reserved: workgroups that never lead (-1 auto); cache_rows: 0 auto."""
        ctx = self._ctx()
        ctx.check(ctx.lib.mplx_set_helpers(ctx.h, int(per_leader), int(reserved), int(cache_rows)))

    def helperStats(self):
        ctx = self._ctx()
        st = (C.c_uint32 * 4)()
        ctx.check(ctx.lib.mplx_helper_stats(ctx.h, st))
        return dict(zip(("cache_rows_used", "queries_done", "helpers_gave_up", "helpers_surplus"), [int(x) for x in st]))

    def setRecord(self, cap):
        ctx = self._ctx()
        ctx.check(ctx.lib.mplx_set_record(ctx.h, int(cap)))
        self._record = int(cap)

    def _ctx(self):
        if self.map_util_ is None:
            raise MplxError("setMapUtil first")
        return self.map_util_.ctx

    def _own_results(self):
        """The device keeps the state space of the context's LAST plan only: refuse to answer from a plan
        another planner object made on the shared context since (instead of returning its state space)."""
        ctx = self._ctx()
        if self._use_lpastar and self._lpa is not None and self._results is not None:
            return ctx  # (an LPA* planner's state space is its own: nothing another planner does can replace it)
        if self._results is None or ctx.lib.mplx_plan_epoch(ctx.h) != getattr(self, "_epoch", -1):
            raise MplxError("the results of this planner's last plan() are gone: another planner sharing the MapUtil planned since")
        return ctx

    def _configure(self, control):
        if self._U is None:
            raise MplxError("setU first")
        ctx = self._ctx()
        # a MapUtil (= one device context) may be shared by several planner objects, like the reference's
        # planner_ / replan_planner_ pair (map_replanner_node.cpp:415,427): re-send the set-up when another
        # planner configured the context since
        owner = getattr(ctx, "cfg_owner", None)  # (a weak reference: the context must not keep its planners -- and through
        if not self._dirty and control == self._control and owner is not None and owner() is self:  # them itself -- alive)
            return
        cfg = _capi.Config()
        cfg.control = control
        cfg.n_u = self._U.shape[0]
        cfg.U = self._U.ctypes.data_as(C.POINTER(C.c_double))
        cfg.dt, cfg.v_max, cfg.a_max, cfg.j_max = self._dt, self._v_max, self._a_max, self._j_max
        cfg.w, cfg.eps = self._w, self._eps
        cfg.tol_pos, cfg.tol_vel, cfg.tol_acc = self._tol
        cfg.t_max = self._t_max
        cfg.max_expand = self._max_num
        cfg.heur_ignore_dynamics = int(self._heur_ignore_dynamics)
        if self._U_yaw is not None:
            cfg.U_yaw = self._U_yaw.ctypes.data_as(C.POINTER(C.c_double))
        cfg.yaw_max, cfg.tol_yaw = self._yaw_max, -1.0
        ctx.check(ctx.lib.mplx_planner_config(ctx.h, C.byref(cfg)))
        ctx.cfg_owner = weakref.ref(self)
        self._control = control
        self._dirty = False

    # ---- planning
    def plan(self, start, goal):
        """bool PlannerBase::plan(start, goal)."""
        if self._U_yaw is not None and not start.use_yaw:
            # (upstream would build yaw-less primitives from the first three components; nothing in the reference does
            # this, and dropping the column silently would plan something else than what was asked for)
            raise MplxError("a 4-component lattice (setU with a yaw column) needs use_yaw start states")
        ctx = self._ctx()
        self._configure(start.control)
        self._apply_aux()
        res = _capi.Result()
        s, g = start.to_c(), goal.to_c()
        if self._use_lpastar:
            lpa = self._lpa_handle()
            if self._record:
                lpa.check(ctx.lib.mplx_lpa_set_record(lpa.h, self._record))
            lpa.check(ctx.lib.mplx_lpa_plan(lpa.h, C.byref(s), C.byref(g), C.byref(res)))
        else:
            ctx.check(ctx.lib.mplx_plan(ctx.h, C.byref(s), C.byref(g), C.byref(res)))
        self._result = res
        self._results = [res]
        self._epoch = ctx.lib.mplx_plan_epoch(ctx.h)
        self.traj_cost_ = res.cost
        if res.status == _capi.PLAN_START_OCCUPIED:
            if self.planner_verbose_:
                print("[PlannerBase] start is not free!")
            return False
        if math.isinf(res.cost):
            if self.planner_verbose_:
                print("[MPPlanner] Cannot find a traj! status", res.status)
            return False
        if res.status != _capi.PLAN_OK:
            # e.g. MPLX_PLAN_TRAJ_TOO_LONG: the goal was reached and the cost is known, but no trajectory came back --
            # never report success with an empty trajectory (a replanner would execute it)
            print(f"\x1b[31m[MPPlanner] plan() failed with status {res.status}: goal reached (cost {res.cost}) but the trajectory "
                  f"has more primitives than the device-side recoverTraj buffer holds\x1b[0m", file=sys.stderr)
            self.traj_cost_ = math.inf
            return False
        return True

    def planBatch(self, starts, goals):
        """Independent queries on the shared map in one launch; returns the list of results."""
        ctx = self._ctx()
        self._configure(starts[0].control)
        self._apply_aux()
        n = len(starts)
        S = (_capi.Waypoint * n)(*[s.to_c() for s in starts])
        G = (_capi.Waypoint * n)(*[g.to_c() for g in goals])
        R = (_capi.Result * n)()
        ctx.check(ctx.lib.mplx_plan_batch(ctx.h, n, S, G, R))
        self._results = [R[i] for i in range(n)]
        self._result = self._results[0]
        self._epoch = ctx.lib.mplx_plan_epoch(ctx.h)
        return self._results

    # ---- streamed batches (include/mplx.h: mplx_plan_batch_submit / _wait, mplx_stream_*)
    def planBatchSubmit(self, starts, goals):
        """First half of planBatch: returns once the batch is launched on the context's stream."""
        ctx = self._ctx()
        self._configure(starts[0].control)
        self._apply_aux()
        n = len(starts)
        S = (_capi.Waypoint * n)(*[s.to_c() for s in starts])
        G = (_capi.Waypoint * n)(*[g.to_c() for g in goals])
        ctx.check(ctx.lib.mplx_plan_batch_submit(ctx.h, n, S, G))
        self._pending_n = n

    def planBatchDone(self):
        ctx = self._ctx()
        rc = ctx.lib.mplx_plan_batch_done(ctx.h)
        if rc < 0:
            ctx.check(rc)
        return rc == 1

    def planBatchWait(self):
        ctx = self._ctx()
        n = self._pending_n
        R = (_capi.Result * n)()
        ctx.check(ctx.lib.mplx_plan_batch_wait(ctx.h, R))
        self._results = [R[i] for i in range(n)]
        self._result = self._results[0]
        self._epoch = ctx.lib.mplx_plan_epoch(ctx.h)
        return self._results

    def setHelperLimit(self, limit):
        """At most `limit` workgroups of a launch stay on as helpers once its query queue is empty (-1: all)."""
        ctx = self._ctx()
        ctx.check(ctx.lib.mplx_set_helper_limit(ctx.h, int(limit)))

    def setPoolRecycling(self, on=True):
        """Batches on the speculative kernels: finished queries hand their pool chunks back (mplx_set_pool_recycling), so setCapacity
        has to cover the concurrently running queries only.  Same results; the batch's state spaces are not kept."""
        ctx = self._ctx()
        ctx.check(ctx.lib.mplx_set_pool_recycling(ctx.h, 1 if on else 0))

    def setDeadline(self, seconds):
        """Launch guard: a search launch older than `seconds` is aborted and the call raises MplxError (MPLX_ERR_TIMEOUT) with
        the workgroups' watch records; <= 0: wait for ever.  Default: none (opt-in; environment MPLX_DEADLINE_S); the clock starts at the launch."""
        ctx = self._ctx()
        ctx.check(ctx.lib.mplx_set_deadline(ctx.h, float(seconds)))

    def _debugHangNextLaunch(self):
        """(tests) the next search launch spins until the deadline aborts it."""
        ctx = self._ctx()
        ctx.check(ctx.lib.mplx_debug_hang_next_launch(ctx.h))

    def releasePools(self):
        """Give the context's device pools back (re-created by the next plan): room for a stream's lanes."""
        ctx = self._ctx()
        ctx.check(ctx.lib.mplx_release_pools(ctx.h))

    def stream(self, depth=2):
        """A PlanStream of `depth` lanes on this planner's map replica and set-up (the planner must be configured:
        plan once, or call configure(control))."""
        return PlanStream(self, depth)

    def configure(self, control):
        self._configure(control)
        self._apply_aux()

    def lastKernelMs(self):
        ctx = self._ctx()
        ms = C.c_float()
        if self._use_lpastar and self._lpa is not None:
            self._lpa.check(ctx.lib.mplx_lpa_last_kernel_ms(self._lpa.h, C.byref(ms)))
        else:
            ctx.check(ctx.lib.mplx_last_kernel_ms(ctx.h, C.byref(ms)))
        return ms.value

    def kernelName(self):
        """Name of the search kernel mplx_plan / mplx_plan_batch launches for the current configuration."""
        ctx = self._ctx()
        return ctx.lib.mplx_kernel_name(ctx.h).decode()

    def queryTiming(self, q=0):
        """(begin_s, end_s, workgroup) of query q on the device clock, relative to the batch start."""
        ctx = self._ctx()
        b, e, s = C.c_double(), C.c_double(), C.c_int32()
        ctx.check(ctx.lib.mplx_result_timing(ctx.h, q, C.byref(b), C.byref(e), C.byref(s)))
        return b.value, e.value, s.value

    def queryCycles(self, q=0):
        ctx = self._ctx()
        cyc = (C.c_uint64 * 10)()
        ctx.check(ctx.lib.mplx_result_cycles(ctx.h, q, cyc))
        return dict(zip(("pop", "expand", "lookup", "evict", "refill", "activate", "commit", "batches", "dep_batches", "cache_hits"), [int(x) for x in cyc[:10]]))

    def querySpeculation(self, q=0):
        """Speculation accounting of query q (mplx_result_speculation): candidates taken, stale entries dropped, units expanded,
        expanded units cut (returned to OPEN and expanded again later)."""
        ctx = self._ctx()
        sp = (C.c_uint64 * 4)()
        ctx.check(ctx.lib.mplx_result_speculation(ctx.h, q, sp))
        return dict(zip(("candidates", "stale", "units_expanded", "units_cut"), [int(x) for x in sp[:4]]))

    # ---- results
    def getTrajCost(self):
        return self.traj_cost_

    def getResult(self, q=0):
        return self._results[q]

    def getTraj(self, q=0):
        ctx = self._own_results()
        res = self._results[q]
        n = res.traj_len if res.status == _capi.PLAN_OK else 0
        if self._use_lpastar and self._lpa is not None:
            n = int(ctx.lib.mplx_lpa_traj_len(self._lpa.h))  # the stored Trajectory of the last SUCCESSFUL plan
        if n <= 0:
            return Trajectory3D([], [], np.zeros(0, dtype=np.int32), res.cost)
        prs = (_capi.Primitive * n)()
        wps = (_capi.Waypoint * (n + 1))()
        act = (C.c_int32 * n)()
        ids = (C.c_int32 * (n + 1))()
        if self._use_lpastar and self._lpa is not None:
            self._lpa.check(ctx.lib.mplx_lpa_result_traj(self._lpa.h, prs, wps, act, ids))
        else:
            ctx.check(ctx.lib.mplx_result_traj(ctx.h, q, prs, wps, act, ids))
        P = [Primitive3D([list(prs[i].c[k]) for k in range(3)], prs[i].t, prs[i].control, list(prs[i].cyaw)) for i in range(n)]
        W = [Waypoint3D.from_c(wps[i]) for i in range(n + 1)]
        tr = Trajectory3D(P, W, np.array(act[:], dtype=np.int32), res.cost)
        tr.node_ids = np.array(ids[:], dtype=np.int32)
        return tr

    def getExpandedIds(self, q=0):
        ctx = self._own_results()
        cap = int(self._results[q].n_expanded)
        ids = np.zeros(max(cap, 1), dtype=np.int32)
        n = C.c_uint32()
        if self._use_lpastar and self._lpa is not None:
            self._lpa.check(ctx.lib.mplx_lpa_result_expanded(self._lpa.h, cap, ids.ctypes.data, C.byref(n)))
        else:
            ctx.check(ctx.lib.mplx_result_expanded(ctx.h, q, cap, ids.ctypes.data, C.byref(n)))
        return ids[:n.value]

    def _nodes(self):
        ctx = self._own_results()
        if self._use_lpastar and self._lpa is not None:
            st = self.lpaStateSpace()
            return st["coords"], st["pos"], st["g"], st["h"], st["closed"], st["opened"]
        n = int(self._result.n_nodes)
        coords = (_capi.Waypoint * max(n, 1))()
        g = np.zeros(n)
        h = np.zeros(n)
        closed = np.zeros(n, dtype=np.int32)
        opened = np.zeros(n, dtype=np.int32)
        if n:
            ctx.check(ctx.lib.mplx_result_nodes(ctx.h, n, coords, g.ctypes.data, h.ctypes.data, closed.ctypes.data, opened.ctypes.data))
        pos = np.array([coords[i].pos[:] for i in range(n)]).reshape(n, 3)
        return coords, pos, g, h, closed, opened

    def lpaStateSpace(self):
        """The LPA* state space as it stands (after a plan, a map edit or a re-rooting): coords / pos / g / rhs / h /
        closed / opened / built per state, and the predecessor entries (child, parent, action, blocked)."""
        lpa, lib = self._lpa, self._lpa.lib
        nn, ne, nb = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        lpa.check(lib.mplx_lpa_counts(lpa.h, C.byref(nn), C.byref(ne), C.byref(nb)))
        n = int(nn.value)
        coords = (_capi.Waypoint * max(n, 1))()
        g, rhs, h = np.zeros(n), np.zeros(n), np.zeros(n)
        closed, opened, built = (np.zeros(n, dtype=np.int32) for _ in range(3))
        if n:
            lpa.check(lib.mplx_lpa_result_nodes(lpa.h, n, coords, g.ctypes.data, rhs.ctypes.data, h.ctypes.data, closed.ctypes.data, opened.ctypes.data, built.ctypes.data))
        m = int(ne.value)
        child, parent, action, blocked = (np.zeros(max(m, 1), dtype=np.int32) for _ in range(4))
        got = C.c_uint64(0)
        lpa.check(lib.mplx_lpa_result_edges(lpa.h, child.ctypes.data, parent.ctypes.data, action.ctypes.data, blocked.ctypes.data, m, C.byref(got)))
        m = int(got.value)
        pos = np.array([coords[i].pos[:] for i in range(n)]).reshape(n, 3)
        return dict(n_nodes=n, n_edges=m, n_blocked_log=int(nb.value), coords=coords, pos=pos, g=g, rhs=rhs, h=h, closed=closed, opened=opened, built=built,
                    child=child[:m], parent=parent[:m], action=action[:m], blocked=blocked[:m])

    def getCloseSet(self):
        _, pos, _, _, closed, _ = self._nodes()
        return pos[closed == 1]

    def getOpenSet(self):
        _, pos, _, _, closed, opened = self._nodes()
        return pos[(opened == 1) & (closed == 0)]

    def getEdges(self):
        """Predecessor lists of the state space: (child, parent, action) int32 arrays, for every node in id
        order its edges in arrival order (StateSpace pred_coord / pred_action_id, poly_map_planner.h:70-86)."""
        ctx = self._own_results()
        if self._use_lpastar and self._lpa is not None:
            st = self.lpaStateSpace()
            return st["child"], st["parent"], st["action"]
        n = int(self._result.n_edges)
        child = np.zeros(max(n, 1), dtype=np.int32); parent = np.zeros(max(n, 1), dtype=np.int32); action = np.zeros(max(n, 1), dtype=np.int32)
        m = C.c_uint64(0)
        ctx.check(ctx.lib.mplx_result_edges(ctx.h, child.ctypes.data, parent.ctypes.data, action.ctypes.data, n, C.byref(m)))
        return child[:m.value], parent[:m.value], action[:m.value]

    def getBlockedEdges(self):
        """(parent id, action) of every successor get_succ emitted with cost inf during the last plan() -- the
        inf-cost pred entries of upstream's state space -- and hm_.size() as upstream counts it."""
        ctx = self._own_results()
        n = C.c_uint64(0)
        tot = C.c_uint64(0)
        ctx.check(ctx.lib.mplx_result_blocked(ctx.h, None, None, 0, C.byref(n), C.byref(tot)))
        parent = np.zeros(max(n.value, 1), dtype=np.int32); action = np.zeros(max(n.value, 1), dtype=np.int32)
        ctx.check(ctx.lib.mplx_result_blocked(ctx.h, parent.ctypes.data, action.ctypes.data, n.value, C.byref(n), C.byref(tot)))
        return parent[:n.value], action[:n.value], int(tot.value)

    def _primitives(self, coords, parent, action):
        prs = []
        for p, a in zip(parent, action):
            w = coords[int(p)]
            st = [w.pos[:], w.vel[:], w.acc[:], w.jrk[:]]
            co = np.zeros((3, 6))
            for ax in range(3):
                co[ax, 5] = st[0][ax]
                if self._control >= VEL: co[ax, 4] = self._U[a][ax] if self._control == VEL else st[1][ax]
                if self._control >= ACC: co[ax, 3] = self._U[a][ax] if self._control == ACC else st[2][ax]
                if self._control >= JRK: co[ax, 2] = self._U[a][ax] if self._control == JRK else st[3][ax]
                if self._control >= SNP: co[ax, 1] = self._U[a][ax]
            prs.append(Primitive3D(co, self._dt, self._control))
        return prs

    def getValidPrimitives(self):
        """PlannerBase::getValidPrimitives: the primitive of every finite-cost pred entry of the state space,
        Primitive(parent state, U[action], dt)."""
        coords, _, _, _, _, _ = self._nodes()
        _, parent, action = self.getEdges()
        return self._primitives(coords, parent, action)

    def getAllPrimitives(self):
        """PlannerBase::getAllPrimitives (poly_map_replanner_node.cpp:184,234): every pred entry of the state
        space, the blocked (cost inf) ones included -- those are re-derived on request (mplx_result_blocked)."""
        coords, _, _, _, _, _ = self._nodes()
        _, parent, action = self.getEdges()
        bp, ba, _ = self.getBlockedEdges()
        return self._primitives(coords, np.concatenate([parent, bp]), np.concatenate([action, ba]))

    def getExpandedNodes(self):
        """Positions in expansion order (env_base::expanded_nodes_); needs setRecord()."""
        ids = self.getExpandedIds(0)
        _, pos, _, _, _, _ = self._nodes()
        return pos[ids]

    # ---- unit entry: env_map::get_succ for many nodes
    def getSuccBatch(self, nodes):
        ctx = self._ctx()
        self._configure(nodes[0].control)
        self._apply_aux()
        K = len(nodes)
        N = (_capi.Waypoint * K)(*[n.to_c() for n in nodes])
        out = (_capi.Succ * (K * self._U.shape[0]))()
        ctx.check(ctx.lib.mplx_expand_batch(ctx.h, K, N, out))
        return out

    def heuristicBatch(self, states, goal):
        ctx = self._ctx()
        self._configure(states[0].control)
        n = len(states)
        S = (_capi.Waypoint * n)(*[s.to_c() for s in states])
        g = goal.to_c()
        h = np.zeros(n)
        isg = np.zeros(n, dtype=np.int32)
        ctx.check(ctx.lib.mplx_heuristic_batch(ctx.h, n, S, C.byref(g), h.ctypes.data, isg.ctypes.data))
        return h, isg


# ---------------------------------------------------------------------------------------------
# 2-D planners (OccMapUtil / OccMapPlanner, used by distance_map_planner_node.cpp:56,146-156):
# the 2-D lattice is the 3-D path with z frozen -- one layer of voxels whose centre plane is z = 0,
# control inputs (ux, uy, 0).  Every z term of the polynomial, key, cost and heuristic arithmetic is
# an exact +0.0, so states, keys, costs and the expansion order equal those of a genuinely 2-D run.
class PlanStream:
    """Host mirror of mplx_stream: several query batches in flight on one map replica.  While the longest queries of
    batch n still run (a query is a serial pop chain on one compute unit), the workgroups of batch n + 1 take the rest
    of the machine.  submit() -> ticket, done(ticket), wait(ticket) -> results (+ the lane's trajectories / timings until
    that lane is submitted to again)."""

    def __init__(self, planner, depth=2):
        self._pl = planner
        self._ctx = planner._ctx()
        self.lib = self._ctx.lib
        h = C.c_void_p()
        self._ctx.check(self.lib.mplx_stream_create(self._ctx.h, int(depth), C.byref(h)))
        self.h = h
        self.depth = int(depth)
        self._n = {}
        self._last_lane = None  # context of the lane last waited for (answers trajectories / timings of that batch)

    def close(self):
        if self.h is not None:
            self.lib.mplx_stream_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise MplxError(f"mplx_stream: {self.lib.mplx_stream_last_error(self.h).decode()} (rc {rc})")
        return rc

    def configure(self, n_slots=0, max_nodes=0, max_edges=0, max_open_log=0, helpers=-1, reserved=-1, cache_rows=0, helper_limit=-1):
        self._check(self.lib.mplx_stream_configure(self.h, n_slots, max_nodes, max_edges, max_open_log, helpers, reserved, cache_rows, helper_limit))

    def submit(self, starts, goals):
        n = len(starts)
        S = (_capi.Waypoint * n)(*[s.to_c() for s in starts])
        G = (_capi.Waypoint * n)(*[g.to_c() for g in goals])
        return self.submit_c(S, G, n)

    def submit_c(self, S, G, n):
        """submit with pre-marshalled ctypes arrays (a steady stream re-uses them)"""
        # setters called on the planner since its last plan reach the parent context now; the C side then hands the
        # set-up to the (idle) lanes -- a streamed batch always plans with what planBatch would plan with
        if self._pl._dirty and self._pl._control is not None:
            self._pl._configure(self._pl._control)
        self._pl._apply_aux()
        t = C.c_int64(-1)
        self._check(self.lib.mplx_stream_submit(self.h, n, S, G, C.byref(t)))
        self._n[t.value] = n
        return t.value

    def done(self, ticket):
        return self._check(self.lib.mplx_stream_done(self.h, int(ticket))) == 1

    def wait(self, ticket):
        n = self._n.pop(int(ticket))
        R = (_capi.Result * n)()
        lane = C.c_void_p()
        self._check(self.lib.mplx_stream_wait(self.h, int(ticket), R, C.byref(lane)))
        self._last_lane = lane
        return [R[i] for i in range(n)]

    def lastKernelMs(self):
        ms = C.c_float()
        self.lib.mplx_last_kernel_ms(self._last_lane, C.byref(ms))
        return ms.value

    def queryTiming(self, q=0):
        b, e, sl = C.c_double(), C.c_double(), C.c_int32()
        self.lib.mplx_result_timing(self._last_lane, q, C.byref(b), C.byref(e), C.byref(sl))
        return b.value, e.value, sl.value

    def trajActions(self, q, traj_len):
        """actions of query q's trajectory in the batch last waited for"""
        act = (C.c_int32 * max(int(traj_len), 1))()
        if traj_len > 0:
            self.lib.mplx_result_traj(self._last_lane, q, None, None, act, None)
        return np.array(act[:int(traj_len)], dtype=np.int32)


class Waypoint2D(Waypoint3D):
    """Waypoint<2>: only the first two components of pos/vel/acc/jrk are meaningful."""

    def __init__(self, control=0):
        super().__init__(control)

    def to_c(self):
        for a in (self.pos, self.vel, self.acc, self.jrk):
            if len(a) == 2:
                a.resize(3, refcheck=False)
            a[2] = 0.0
        return super().to_c()


class OccMapUtil(VoxelMapUtil):
    """MapUtil<2>: setMap(ori (2), dim (2), data, res)."""

    def setMap(self, ori, dim, data, res):
        res = float(res)
        super().setMap((float(ori[0]), float(ori[1]), -0.5 * res), (int(dim[0]), int(dim[1]), 1), data, res)

    def getDim(self):
        return super().getDim()[:2]

    def getOrigin(self):
        return super().getOrigin()[:2]

    def query(self, pts):
        p = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 2)
        cells, st = super().query(np.hstack([p, np.zeros((p.shape[0], 1))]))
        return cells[:, :2], st


class OccMapPlanner(VoxelMapPlanner):
    """PlannerBase<2, Waypoint2D> + MapPlanner<2>."""

    def setU(self, U):
        U = np.asarray(U, dtype=np.float64)
        U = U.reshape(-1, U.shape[-1])
        if U.shape[1] == 2:
            U = np.hstack([U, np.zeros((U.shape[0], 1))])
        if np.any(U[:, 2] != 0):
            raise ValueError("OccMapPlanner needs planar control inputs")
        super().setU(U)
