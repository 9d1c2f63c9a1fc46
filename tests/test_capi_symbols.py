"""CPU: the C-ABI library loads and exports every symbol include/mplx.h declares; without a GPU
every compute path fails loudly (no fallback)."""
import ctypes as C
import os
import re

import pytest

from mpl_ros_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "mplx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mplx_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_list_agree():
    assert declared_symbols() == sorted(_capi.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = _capi.load()
    for name in declared_symbols():
        assert getattr(lib, name) is not None
    assert lib.mplx_version().startswith(b"mplx")


def test_struct_layouts_match_the_header():
    assert C.sizeof(_capi.Waypoint) == 14 * 8 + 8
    assert C.sizeof(_capi.Primitive) == 18 * 8 + 8 + 8 + 6 * 8  # c[3][6], t, control + pad, cyaw[6]
    assert C.sizeof(_capi.Config) == 8 + 8 + 10 * 8 + 8 + 8 + 2 * 8  # control n_u | U | 10 doubles | 2 ints | U_yaw | yaw_max tol_yaw
    assert C.sizeof(_capi.Succ) == C.sizeof(_capi.Waypoint) + 8 + 8 + 48 + 8
    assert C.sizeof(_capi.Result) == 16 + 13 * 8


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _capi.load()
    h = C.c_void_p()
    assert lib.mplx_ctx_create(0, C.byref(h)) == _capi.ERR_HIP
    assert b"HIP device" in lib.mplx_last_error(None)
    from mpl_ros_amd.planner import VoxelMapUtil
    with pytest.raises(_capi.MplxError):
        VoxelMapUtil()


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "mpl_ros_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".inl")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("no CPU fallback", ""), f


def test_reference_api_that_needs_no_device_behaves():
    """Without a state space the LPA* update calls change nothing; the potential-field setters only store (the device is
    touched by updatePotentialMap / plan); a gradient weight other than the reference's 0 is refused, never ignored."""
    import pytest
    from mpl_ros_amd._capi import MplxError
    from mpl_ros_amd.planner import VoxelMapPlanner
    pl = VoxelMapPlanner(False)
    pl.setLPAstar(False)  # the default
    pl.setLPAstar(True)
    assert not pl.initialized()
    assert pl.updateBlockedNodes([]) == 0 and pl.updateClearedNodes([]) == 0 and pl.getSubStateSpace(1) is None
    pl.setSearchRadius([0.5, 0.5]); pl.setPotentialRadius([1.5, 1.5]); pl.setPotentialWeight(10); pl.setGradientWeight(0)
    with pytest.raises(MplxError):
        pl.setGradientWeight(0.5)
    with pytest.raises(MplxError):
        pl.updatePotentialMap([0, 0])  # no MapUtil yet


def test_streamed_batch_entry_points_refuse_null_handles_without_touching_a_device():
    """include/mplx.h "streamed batches": every entry point answers MPLX_ERR_ARG (or an empty value) for a null handle."""
    lib = _capi.load()
    t = C.c_int64(-1)
    out = C.c_void_p()
    assert lib.mplx_stream_create(None, 2, C.byref(out)) == _capi.ERR_ARG and not out.value
    assert lib.mplx_stream_depth(None) == 0
    assert lib.mplx_stream_last_error(None) == b""
    assert lib.mplx_stream_submit(None, 1, None, None, C.byref(t)) == _capi.ERR_ARG and t.value == -1
    assert lib.mplx_stream_done(None, 0) == _capi.ERR_ARG
    assert lib.mplx_stream_wait(None, 0, None, None) == _capi.ERR_ARG
    assert lib.mplx_stream_configure(None, 1, 1, 1, 1, -1, 0, 0, -1) == _capi.ERR_ARG
    lib.mplx_stream_destroy(None)  # (a no-op)
    assert lib.mplx_plan_batch_wait(None, None) == _capi.ERR_ARG
    assert lib.mplx_plan_batch_done(None) == _capi.ERR_ARG
    assert lib.mplx_set_helper_limit(None, 4) == _capi.ERR_ARG
    assert lib.mplx_release_pools(None) == _capi.ERR_ARG
