"""Extracts the VoxelMap grid from the reference's one surviving map fixture
(/root/reference/mpl_test_node/maps/skir/skir.bag) into tests/golden/skir_map.npz.

Run in the build container only (the GPU box has no /root/reference).  The bag is an uncompressed
rosbag v2.0; the planning_ros_msgs/VoxelMap message layout is Header, float32 resolution,
3 x float64 origin, 3 x float64 dim, uint32 n, n x int8 (planning_ros_msgs/msg/VoxelMap.msg:1-12).
"""
import hashlib
import os
import struct

import numpy as np

BAG = "/root/reference/mpl_test_node/maps/skir/skir.bag"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "skir_map.npz")


def find_voxel_map(buf):
    # locate the serialized message: frame_id-length + resolution + origin + dim + n, with
    # n == dimx*dimy*dimz and n bytes following.  Scan for a plausible (dim, n) pair.
    for off in range(0, len(buf) - 64):
        res = struct.unpack_from("<f", buf, off)[0]
        if not (0.0999 < res < 0.1001):
            continue
        ox, oy, oz, dx, dy, dz = struct.unpack_from("<6d", buf, off + 4)
        n = struct.unpack_from("<I", buf, off + 52)[0]
        if dx > 0 and dy > 0 and dz > 0 and dx == int(dx) and dy == int(dy) and dz == int(dz) \
                and n == int(dx) * int(dy) * int(dz) and off + 56 + n <= len(buf):
            return off, res, (ox, oy, oz), (int(dx), int(dy), int(dz)), n
    raise RuntimeError("VoxelMap not found")


def main():
    buf = open(BAG, "rb").read()
    assert buf.startswith(b"#ROSBAG V2.0")
    off, res, origin, dim, n = find_voxel_map(buf)
    data = np.frombuffer(buf, dtype=np.int8, count=n, offset=off + 56).copy()
    sha = hashlib.sha256(data.tobytes()).hexdigest()
    print("offset", off + 56, "res", res, "origin", origin, "dim", dim, "n", n, "sha256", sha)
    vals, cnt = np.unique(data, return_counts=True)
    print(dict(zip(vals.tolist(), cnt.tolist())))
    grid = data.reshape(dim[2], dim[1], dim[0])  # x fastest: idx = x + dx*y + dx*dy*z
    np.savez_compressed(OUT, grid=grid, origin=np.array(origin), res=np.float64(res), sha256=sha)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
