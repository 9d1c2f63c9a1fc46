"""Timing of the MapUtil kernels on the 512^3 benchmark map (GPU box): dilate, getCloud, brick pack."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpl_ros_amd import mapgen
from mpl_ros_amd.planner import VoxelMapUtil

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
grid, origin, res, start, goal, rng = mapgen.benchmark_map(n)
mu = VoxelMapUtil()
dz, dy, dx = grid.shape
t = time.perf_counter(); mu.setMap(origin, (dx, dy, dz), grid.ravel(), res); t_set = time.perf_counter() - t
offs = [(x, y, z) for x in (-1, 0, 1) for y in (-1, 0, 1) for z in (-1, 0, 1) if (x, y, z) != (0, 0, 0)]
for it in range(3):
    t = time.perf_counter(); mu.dilate(offs); dt = time.perf_counter() - t
    print(f"dilate 26 offsets on {n}^3: {dt * 1e3:.2f} ms wall (kernel + bitmap rebuild + sync) -> {2 * n**3 / dt / 1e9:.1f} GB/s of 2 B/voxel")
t = time.perf_counter(); c = mu.getCloud(); dt = time.perf_counter() - t
print(f"getCloud: {len(c)} points in {dt * 1e3:.1f} ms wall (count + scan + write + D2H of {c.nbytes / 1e6:.0f} MB)")
print(f"setMap (H2D 128 MiB + brick pack): {t_set * 1e3:.1f} ms")

# ---- VoxelGrid ingest: cloud -> grid -> planner map, all on the device
from mpl_ros_amd.voxel_grid import VoxelGrid
rng = np.random.default_rng(0)
ext = n * 0.1
G = VoxelGrid((0.0, 0.0, 0.0), (ext, ext, ext), 0.1)
npts = 10_000_000
pts = rng.uniform(0.0, ext, (npts, 3))
for it in range(2):
    t = time.perf_counter(); G.addCloud(pts); dt = time.perf_counter() - t
    print(f"VoxelGrid.addCloud {npts} points: {dt * 1e3:.1f} ms wall incl. H2D of {pts.nbytes / 1e6:.0f} MB -> {npts / dt / 1e6:.0f} Mpoints/s")
ns = [(x, y, 0) for x in (-1, 0, 1) for y in (-1, 0, 1)]
t = time.perf_counter(); new = G.addCloud(pts[:1_000_000] + 0.05, ns); dt = time.perf_counter() - t
print(f"VoxelGrid.addCloud(pts, ns) 1M points x 9 neighbours: {dt * 1e3:.1f} ms, {len(new)} newly inflated cells (ordered)")
t = time.perf_counter(); G.setMapUtil(mu); dt = time.perf_counter() - t
print(f"VoxelGrid -> MapUtil device to device (getMap transform + bitmap rebuild): {dt * 1e3:.2f} ms for {n}^3")
t = time.perf_counter(); G.decay(); dt = time.perf_counter() - t
print(f"VoxelGrid.decay: {dt * 1e3:.2f} ms -> {4 * n**3 / dt / 1e9:.0f} GB/s (2 grids, read + write)")
