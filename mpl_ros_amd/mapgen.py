"""Synthetic voxel maps, control lattices and query sets for the benchmark configs.

Host-side helpers only (numpy): the map they produce is handed to MapUtil.setMap exactly like a
map read from a VoxelMap message would be.
  * random-box maps: SURVEY.md 8(d) C2/C3 -- SplitMix64, seed 20250620, boxes with uniform centres and
    edge lengths uniform in [2,12] voxels until occupancy >= 10 %, 5-voxel free bubbles at start/goal.
  * control lattice: same nested accumulate-by-du loops as the reference driver
    (mpl_test_node/src/map_planner_node.cpp:108-139), generated once on the host and passed as data.
"""
import numpy as np

_M64 = (1 << 64) - 1
VAL_FREE, VAL_OCC, VAL_UNKNOWN = 0, 100, -1  # planning_ros_utils/include/planning_ros_utils/voxel_grid.h:43-45


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & _M64

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & _M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def below(self, n):
        return self.next() % n

    def uniform(self):
        return (self.next() >> 11) * (1.0 / (1 << 53))


def random_box_map(dim, seed=20250620, occupancy=0.10, edge=(2, 12), rng=None):
    """int8 grid of shape (dz, dy, dx) (x fastest in memory) with random occupied boxes."""
    dx, dy, dz = dim
    rng = rng or SplitMix64(seed)
    grid = np.zeros((dz, dy, dx), dtype=np.int8)
    target = int(occupancy * grid.size)
    filled = 0
    span = edge[1] - edge[0] + 1
    while filled < target:
        c = (rng.below(dx), rng.below(dy), rng.below(dz))
        e = (edge[0] + rng.below(span), edge[0] + rng.below(span), edge[0] + rng.below(span))
        lo = [max(0, c[i] - e[i] // 2) for i in range(3)]
        hi = [min(dim[i], lo[i] + e[i]) for i in range(3)]
        sub = grid[lo[2]:hi[2], lo[1]:hi[1], lo[0]:hi[0]]
        filled += sub.size - int(np.count_nonzero(sub))
        sub[...] = VAL_OCC
    return grid, rng


def float_to_cell(pt, origin, res):
    """MapUtil::floatToInt convention: round((p - origin)/res - 0.5) (half away from zero)."""
    v = (np.asarray(pt, dtype=np.float64) - np.asarray(origin, dtype=np.float64)) / res - 0.5
    return np.where(v >= 0, np.floor(v + 0.5), np.ceil(v - 0.5)).astype(np.int64)


def carve_bubble(grid, pt, origin, res, radius=5):
    c = float_to_cell(pt, origin, res)
    dz, dy, dx = grid.shape
    lo = [max(0, int(c[i]) - radius) for i in range(3)]
    hi = [min((dx, dy, dz)[i], int(c[i]) + radius + 1) for i in range(3)]
    grid[lo[2]:hi[2], lo[1]:hi[1], lo[0]:hi[0]] = VAL_FREE


def control_lattice(u=1.0, num=1, use_3d=True, u_yaw=None):
    """U as an (nU, 3) float64 array; loops accumulate `dx += du` like the reference driver.
    u_yaw: the use_yaw lattices (map_planner_node.cpp:119-139): (nU, 4), yaw rates -u_yaw, 0, u_yaw innermost."""
    du = u / num
    out = []

    def emit(x, y, z):
        if u_yaw is None:
            out.append((x, y, z))
            return
        q = -u_yaw
        while q <= u_yaw:
            out.append((x, y, z, q))
            q += u_yaw

    x = -u
    while x <= u:
        y = -u
        while y <= u:
            if use_3d:
                z = -u
                while z <= u:
                    emit(x, y, z)
                    z += du
            else:
                emit(x, y, 0.0)
            y += du
        x += du
    return np.array(out, dtype=np.float64)


def benchmark_map(n, seed=20250620, res=0.1):
    """C2 (n=256) / C3 (n=512) map with start/goal bubbles; returns grid, origin, res, start, goal."""
    origin = (0.0, 0.0, 0.0)
    grid, rng = random_box_map((n, n, n), seed=seed)
    start = (2.05, 2.05, 2.05)
    g = {256: 23.55, 512: 49.15}.get(n, round((n - 20) * res, 2) + 0.05)
    goal = (g, g, g)
    carve_bubble(grid, start, origin, res)
    carve_bubble(grid, goal, origin, res)
    return grid, origin, res, start, goal, rng


def random_queries(grid, origin, res, nq, rng, min_dist=10.0):
    """nq (start, goal) pairs on free cell centres, at least min_dist metres apart (C4)."""
    dz, dy, dx = grid.shape
    out = []

    def draw():
        while True:
            c = (rng.below(dx), rng.below(dy), rng.below(dz))
            if grid[c[2], c[1], c[0]] == VAL_FREE:
                return tuple((c[i] + 0.5) * res + origin[i] for i in range(3))

    while len(out) < nq:
        s, g = draw(), draw()
        if sum((s[i] - g[i]) ** 2 for i in range(3)) ** 0.5 >= min_dist:
            out.append((s, g))
    return out


C4_SEED = 20250620


def c4_queries(grid, origin, res, nq, rank=0, min_dist=10.0):
    """BASELINE.md C4 query stream of `rank` (bench.py, tests/test_gpu_scale.py): nq pairs of free cell centres
    at least min_dist apart, drawn from SplitMix64(C4_SEED + 7919 (rank + 1))."""
    return random_queries(grid, origin, res, nq, SplitMix64(C4_SEED + 7919 * (rank + 1)), min_dist=min_dist)


def c4_pools(control_is_jrk, nq, max_expand, single=False, per_q=0):
    """Pool sizes (states, predecessor records, OPEN-log entries) bench.py and the scale tests use for a
    C4 batch of nq queries (mean states per query; the pools are shared by the batch)."""
    if not control_is_jrk:
        per_q = per_q or 450_000
        return dict(nodes=per_q * nq, edges=per_q * nq * 9 // 2, log=per_q * nq * 5 // 4)
    per_q = per_q or max(1 << 16, max_expand * 24)
    return dict(nodes=per_q * nq, edges=per_q * nq * 7 // 2, log=per_q * nq * 3 // 2)
