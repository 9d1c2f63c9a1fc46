"""The bag reader against the reference's surviving fixtures.  Needs /root/reference (build container
only); the GPU box and CI without it skip -- the committed tests/golden/skir_map.npz is what travels."""
import hashlib
import os
import struct

import numpy as np
import pytest

from mpl_ros_amd import rosbag_io

REF = "/root/reference/mpl_test_node/maps"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record(fields, data):
    h = b"".join(struct.pack("<I", len(k) + 1 + len(v)) + k + b"=" + v for k, v in fields)
    return struct.pack("<I", len(h)) + h + struct.pack("<I", len(data)) + data


def test_reader_on_a_hand_built_bag(tmp_path):
    # one connection + one VoxelMap message inside an uncompressed chunk, written field by field
    grid = (np.arange(24) % 3 == 0).astype(np.int8) * 100
    msg = (struct.pack("<III", 7, 1, 2) + struct.pack("<I", 3) + b"map" + struct.pack("<f", 0.25) + struct.pack("<6d", 1.0, 2.0, 3.0, 4.0, 3.0, 2.0)
           + struct.pack("<I", 24) + grid.tobytes())
    conn_data = b"".join(struct.pack("<I", len(x)) + x for x in (b"topic=/voxel_map", b"type=planning_ros_msgs/VoxelMap"))
    inner = _record([(b"op", bytes([7])), (b"conn", struct.pack("<I", 0)), (b"topic", b"/voxel_map")], conn_data)
    inner += _record([(b"op", bytes([2])), (b"conn", struct.pack("<I", 0)), (b"time", struct.pack("<II", 5, 6))], msg)
    bag = b"#ROSBAG V2.0\n" + _record([(b"op", bytes([3])), (b"index_pos", struct.pack("<Q", 0)), (b"conn_count", struct.pack("<I", 1)), (b"chunk_count", struct.pack("<I", 1))], b" " * 16)
    bag += _record([(b"op", bytes([5])), (b"compression", b"none"), (b"size", struct.pack("<I", len(inner)))], inner)
    p = tmp_path / "t.bag"
    p.write_bytes(bag)
    m = rosbag_io.read_bag(str(p), "voxel_map")[-1]
    assert m["header"]["frame_id"] == "map" and m["resolution"] == np.float32(0.25)
    assert m["origin"] == (1.0, 2.0, 3.0) and m["dim"] == (4, 3, 2) and np.array_equal(m["data"], grid)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference fixtures not present")
def test_skir_bag_matches_the_committed_fixture():
    m = rosbag_io.read_bag(os.path.join(REF, "skir", "skir.bag"), "voxel_map")[-1]
    d = np.load(os.path.join(ROOT, "tests", "golden", "skir_map.npz"))
    assert m["dim"] == tuple(d["grid"].shape[::-1]) and np.allclose(m["origin"], d["origin"]) and float(m["resolution"]) == pytest.approx(float(d["res"]))
    assert hashlib.sha256(m["data"].tobytes()).hexdigest() == str(d["sha256"])
    assert np.array_equal(m["data"].reshape(d["grid"].shape), d["grid"])
