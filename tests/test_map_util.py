"""MapUtil helpers next to the search: dilate, cell queries, rayTrace, clouds.
CPU part pins the oracle restatement on hand-checked cases; GPU part compares the HIP path with it."""
import numpy as np
import pytest

from mpl_ros_amd import mapgen
from oracle import orc
from tests import util


def _oracle(grid, origin, res):
    P = orc.Planner()
    P.set_map(grid, origin, res)
    return P


def _disc_offsets(rn, hn):
    ns = []
    for nx in range(-rn, rn + 1):
        for ny in range(-rn, rn + 1):
            if np.hypot(nx, ny) > rn:
                continue
            for nz in range(-hn, hn + 1):
                if nx == 0 and ny == 0 and nz == 0:
                    continue
                ns.append((nx, ny, nz))
    return np.array(ns, dtype=np.int32)


def test_oracle_dilate_does_not_cascade_and_clips_at_the_border():
    grid = np.zeros((4, 5, 6), dtype=np.int8)  # [z][y][x]
    grid[1, 2, 0] = 100
    grid[3, 4, 5] = -1  # unknown stays unknown unless a neighbour dilates into it
    P = _oracle(grid, (0, 0, 0), 0.5)
    P.dilate([(1, 0, 0), (-1, 0, 0), (0, 0, 1)])
    g = P.get_map()
    want = grid.copy()
    want[1, 2, 1] = 100   # +x
    want[2, 2, 0] = 100   # +z;  -x falls outside the map
    assert np.array_equal(g, want)


def test_oracle_ray_trace_and_cloud_small_cases():
    grid = np.zeros((2, 3, 4), dtype=np.int8)
    grid[0, 1, 2] = 100
    grid[1, 0, 3] = 50
    grid[1, 2, 0] = -1
    P = _oracle(grid, (1.0, 2.0, 3.0), 0.5)
    occ = P.cloud(0)  # x outermost: (2,1,0) before (3,0,1)
    assert np.allclose(occ, [[1.0 + 2.5 * 0.5, 2.0 + 1.5 * 0.5, 3.0 + 0.5 * 0.5], [1.0 + 3.5 * 0.5, 2.0 + 0.5 * 0.5, 3.0 + 1.5 * 0.5]])
    assert len(P.cloud(2)) == 1 and len(P.cloud(1)) == 24 - 3
    cells = P.ray_trace((1.1, 2.1, 3.1), (2.9, 2.1, 3.1))  # along +x through cells 0..3, end points excluded
    assert cells[:, 1].tolist() == [0] * len(cells) and cells[:, 2].tolist() == [0] * len(cells)
    assert cells[:, 0].tolist() == sorted(set(cells[:, 0].tolist())) and set(cells[:, 0].tolist()) <= {0, 1, 2, 3}
    assert P.cell_state((2, 1, 0)) == 1 and P.cell_state((0, 2, 1)) == 2 and P.cell_state((0, 0, 0)) == 0 and P.cell_state((4, 0, 0)) == 3
    assert len(P.ray_trace((1.1, 2.1, 3.1), (1.1, 2.1, 3.1))) == 0  # zero-length ray


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 5])
def test_map_util_helpers_match_oracle(seed):
    from mpl_ros_amd.planner import VoxelMapUtil
    grid, origin, res = util.small_map(64, seed=seed, occupancy=0.08)
    rng = np.random.default_rng(seed)
    unk = rng.random(grid.shape) < 0.05
    grid = grid.copy()
    grid[unk & (grid == 0)] = -1
    P = _oracle(grid, origin, res)
    mu = VoxelMapUtil()
    dz, dy, dx = grid.shape
    mu.setMap(origin, (dx, dy, dz), grid.ravel(), res)
    # clouds before dilation: same points, same order
    for which, fn in ((0, mu.getCloud), (1, mu.getFreeCloud), (2, mu.getUnknownCloud)):
        a, b = fn(), P.cloud(which)
        assert a.shape == b.shape and np.array_equal(a, b)
    # dilate with the reference node's disc (map_planner_node.cpp:75-83) and a 3-D variant
    for offs in (_disc_offsets(2, 0), _disc_offsets(1, 1)):
        mu.dilate(offs)
        P.dilate(offs)
        assert np.array_equal(mu.getMap().reshape(grid.shape), P.get_map())
    # cells
    cells = rng.integers(-2, 66, size=(500, 3)).astype(np.int32)
    st = mu.cellStates(cells)
    assert st.tolist() == [P.cell_state(c) for c in cells]
    # rays, some leaving the map
    for _ in range(40):
        a = rng.uniform(0.0, 6.4, 3); b = rng.uniform(-1.0, 7.4, 3)
        assert np.array_equal(mu.rayTrace(a, b), P.ray_trace(a, b))
    # a plan on the dilated map still agrees with the oracle (the bitmap was rebuilt)
    cells0, st0 = mu.query(rng.uniform(0.2, 6.2, (200, 3)))
    assert st0.tolist() == [P.cell_state(c) for c in cells0]
