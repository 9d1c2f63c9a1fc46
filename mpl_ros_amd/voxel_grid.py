"""Host mirror of the reference's mapper `VoxelGrid` (planning_ros_utils/include/planning_ros_utils/voxel_grid.h)
over the device-resident grid of libmplx.so.  Same method names and meanings; `getMap()` returns the
fields of planning_ros_msgs/VoxelMap (origin, dim, resolution, data x-fastest) as a dict, and
`setMapUtil(map_util)` is the device-to-device form of `setMap(map_util, voxel_mapper_->getMap())`
(map_replanner_node.cpp:186-188)."""
import ctypes as C

import numpy as np

from . import _capi


class VoxelGrid:
    def __init__(self, origin, dim, res, device=0):
        self.lib = _capi.load()
        self.h = C.c_void_p()
        o = (C.c_double * 3)(*[float(v) for v in origin]); d = (C.c_double * 3)(*[float(v) for v in dim])
        rc = self.lib.mplx_grid_create(device, o, d, float(res), C.byref(self.h))
        if rc != 0:
            raise _capi.MplxError(f"mplx_grid_create: {self.lib.mplx_grid_last_error(None).decode()}")

    def __del__(self):
        try:
            if self.h:
                self.lib.mplx_grid_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise _capi.MplxError(self.lib.mplx_grid_last_error(self.h).decode())

    def info(self):
        dim = (C.c_int32 * 3)(); ori = (C.c_double * 3)(); res = C.c_float()
        self._check(self.lib.mplx_grid_info(self.h, dim, ori, C.byref(res)))
        return tuple(dim), tuple(ori), res.value

    def allocate(self, new_dim_d, new_ori_d):
        ch = C.c_int(0)
        d = (C.c_double * 3)(*[float(v) for v in new_dim_d]); o = (C.c_double * 3)(*[float(v) for v in new_ori_d])
        self._check(self.lib.mplx_grid_allocate(self.h, d, o, C.byref(ch)))
        return bool(ch.value)

    def clear(self, nx=None, ny=None):
        if nx is None:
            self._check(self.lib.mplx_grid_clear(self.h))
        else:
            self._check(self.lib.mplx_grid_clear_column(self.h, int(nx), int(ny)))

    def fill(self, nx, ny, nz=None):
        if nz is None:
            self._check(self.lib.mplx_grid_fill_column(self.h, int(nx), int(ny)))
        else:
            self._check(self.lib.mplx_grid_fill_cell(self.h, int(nx), int(ny), int(nz)))

    def addCloud(self, pts, ns=None):
        p = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 3)
        if ns is None:
            self._check(self.lib.mplx_grid_add_cloud(self.h, p.shape[0], p.ctypes.data))
            return None
        o = np.ascontiguousarray(ns, dtype=np.int32).reshape(-1, 3)
        cap = max(p.shape[0] * o.shape[0], 1)
        out = np.empty((cap, 3), dtype=np.int32)
        n = C.c_int(0)
        self._check(self.lib.mplx_grid_add_cloud_inflate(self.h, p.shape[0], p.ctypes.data, o.shape[0], o.ctypes.data, out.ctypes.data, cap, C.byref(n)))
        return out[:n.value]

    def decay(self):
        self._check(self.lib.mplx_grid_decay(self.h))

    def _map(self, inflated):
        dim, ori, res = self.info()
        data = np.empty(dim[0] * dim[1] * dim[2], dtype=np.int8)
        self._check(self.lib.mplx_grid_get_map(self.h, inflated, data.ctypes.data))
        return {"origin": ori, "dim": dim, "resolution": res, "data": data}

    def getMap(self):
        return self._map(0)

    def getInflatedMap(self):
        return self._map(1)

    def getCloud(self):
        n = C.c_uint64(0)
        self._check(self.lib.mplx_grid_get_cloud(self.h, None, 0, C.byref(n)))
        out = np.empty((max(n.value, 1), 3), dtype=np.float64)
        self._check(self.lib.mplx_grid_get_cloud(self.h, out.ctypes.data, n.value, C.byref(n)))
        return out[:n.value]

    def setMapUtil(self, map_util, inflated=False):
        """getMap() into a VoxelMapUtil without leaving the device"""
        self._check(self.lib.mplx_grid_to_map(self.h, 1 if inflated else 0, map_util.ctx.h))
        map_util._dim = tuple(int(x) for x in self.info()[0])
