"""Regenerates BASELINE config 1 (ii): the reference's `simple` voxel map.

mpl_test_node/maps/simple/simple.bag is a missing blob (.MISSING_LARGE_BLOBS:3); the mesh it was made from,
maps/simple/simple.stl, is in the tree, and so is the recipe: launch/map_generator/samples/mesh_to_map.launch.simple
runs `mesh_sampling` (planning_ros_utils/src/mapping_utils/mesh_sampling.cpp: n_samples 1 000 000 area-weighted
surface samples, then a PCL VoxelGrid filter of leaf 0.05) into `cloud_to_map` (cloud_to_map.cpp: VoxelGrid(origin
(0,0,0), range (18,18,2), res 0.1).addCloud -> getMap).  This script follows that recipe with the VoxelGrid restatement
of oracle/ (pinned against the reference's own voxel_grid.cpp, tests/test_voxel_grid.py).

APPROXIMATE REGENERATION: the sampler draws from C rand() upstream (unseeded, libc-specific); here a seeded numpy
generator.  A surface voxel is hit by ~40 samples, so the occupancy grid is insensitive to the draw, but it is not
claimed to equal the lost simple.bag bit for bit.  Run in the build container (needs /root/reference):

    python tests/golden/make_simple_fixture.py      -> tests/golden/simple_map.npz, tests/golden/simple_plan.json
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
STL = "/root/reference/mpl_test_node/maps/simple/simple.stl"


def read_ascii_stl(path):
    tris, cur = [], []
    for line in open(path):
        t = line.split()
        if t and t[0] == "vertex":
            cur.append([float(t[1]), float(t[2]), float(t[3])])
            if len(cur) == 3:
                tris.append(cur)
                cur = []
    return np.array(tris, dtype=np.float64)  # (n, 3 vertices, 3)


def sample_surface(tris, n, seed):
    """mesh_sampling.cpp:76-129: triangle by cumulative area (lower_bound on a float draw), point by
    (1 - sqrt r1) A + sqrt r1 (1 - r2) B + sqrt r1 r2 C in float32."""
    a, b, c = tris[:, 0], tris[:, 1], tris[:, 2]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    cum = np.cumsum(area)
    rng = np.random.Generator(np.random.PCG64(seed))
    r = (rng.random(n) * cum[-1]).astype(np.float32)
    el = np.minimum(np.searchsorted(cum, r.astype(np.float64), side="left"), len(cum) - 1)
    r1 = rng.random(n).astype(np.float32)
    r2 = rng.random(n).astype(np.float32)
    s = np.sqrt(r1)
    A, B, C_ = a[el].astype(np.float32), b[el].astype(np.float32), c[el].astype(np.float32)
    one_s, one_r2 = (1 - s)[:, None], (1 - r2)[:, None]
    return s[:, None] * (r2[:, None] * C_ + B * one_r2) + A * one_s  # float32 (n, 3)


def leaf_filter(pts, leaf):
    """pcl::VoxelGrid with leaf size `leaf`: the centroid of the points of every occupied leaf."""
    inv = np.float32(1.0 / leaf)
    idx = np.floor(pts * inv).astype(np.int64)
    idx -= idx.min(axis=0)
    dims = idx.max(axis=0) + 1
    lin = idx[:, 0] + dims[0] * (idx[:, 1] + dims[1] * idx[:, 2])
    order = np.argsort(lin, kind="stable")
    lin, p = lin[order], pts[order].astype(np.float64)
    first = np.flatnonzero(np.r_[True, lin[1:] != lin[:-1]])
    cnt = np.diff(np.r_[first, len(lin)])
    return (np.add.reduceat(p, first, axis=0) / cnt[:, None]).astype(np.float32)


def main():
    from mpl_ros_amd import mapgen
    from oracle import orc
    tris = read_ascii_stl(STL)
    assert tris.shape == (80, 3, 3), tris.shape  # 80 facets
    cloud = leaf_filter(sample_surface(tris, 1_000_000, seed=20250620), 0.05)
    G = orc.Grid((0.0, 0.0, 0.0), (18.0, 18.0, 2.0), np.float32(0.1))  # cloud_to_map.cpp + mesh_to_map.launch.simple:21-27
    G.add_cloud(cloud.astype(np.float64))
    dim, ori, res = G.info()
    grid = G.get_map().reshape(dim[2], dim[1], dim[0])
    np.savez_compressed(os.path.join(HERE, "simple_map.npz"), grid=grid, origin=np.array(ori), res=np.float64(res))
    print("simple map", dim, ori, res, "occupied", int((grid > 0).sum()), "cloud", len(cloud))
    # launch/map_planner_node/test.launch:16-33: start (14.5, 4.5, 0.05) at rest -> goal (2.4, 16.6, 0.05), v_max 2, a_max 1,
    # yaw_max 0.5 (use_yaw false), u 1, dt 1, use_3d false (nU = 9), tol 0.5
    U = mapgen.control_lattice(1.0, 1, False)
    P = orc.Planner()
    P.set_map(grid, ori, res)
    P.free_unknown()
    P.set_config(orc.ACC, U, v_max=2.0, a_max=1.0, tol_pos=0.5)
    st = P.plan(orc.waypoint((14.5, 4.5, 0.05)), orc.waypoint((2.4, 16.6, 0.05)))
    ids, _ = P.expanded()
    tr = P.traj()
    h = 0
    for i in ids:
        h = (h * 0x100000001B3 + (int(i) + 1)) & ((1 << 64) - 1)
    plan = dict(name="simple_launch_query", status=int(st), n_expanded=len(ids), expand_hash=str(h), cost=P.traj_cost,
                actions=tr["actions"].tolist(), node_ids=tr["node_ids"].tolist(),
                waypoints=[orc.wp_state(w, orc.ACC).tolist() for w in tr["wps"]], counters=P.counters(), n_closed=P.num_closed())
    json.dump(plan, open(os.path.join(HERE, "simple_plan.json"), "w"), indent=1)
    print("plan status", st, "expanded", len(ids), "cost", P.traj_cost, "primitives", tr["n"])


if __name__ == "__main__":
    main()
