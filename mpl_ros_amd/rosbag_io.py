"""Reader for the map fixtures of the reference: rosbag v2.0 files holding planning_ros_msgs/VoxelMap and
sensor_msgs/PointCloud messages (read by `read_bag<T>(file, topic, 0).back()` at map_planner_node.cpp:63-64
and map_replanner_node.cpp:319-322).  Pure host IO, no ROS: the bag container (records with
`<field_len><name>=<value>` headers; chunks, compression none or bz2) and the two message layouts
(planning_ros_msgs/msg/VoxelMap.msg:1-12; sensor_msgs/PointCloud = Header, Point32[] points,
ChannelFloat32[] channels) are parsed directly."""
import bz2
import struct

import numpy as np

OP_MSG, OP_BAG_HEADER, OP_INDEX, OP_CHUNK, OP_CHUNK_INFO, OP_CONNECTION = 2, 3, 4, 5, 6, 7


def _fields(buf):
    out, i = {}, 0
    while i < len(buf):
        (n,) = struct.unpack_from("<I", buf, i)
        f = buf[i + 4:i + 4 + n]
        k, _, v = f.partition(b"=")
        out[k.decode()] = v
        i += 4 + n
    return out


def _records(buf, i=0, end=None):
    end = len(buf) if end is None else end
    while i + 8 <= end:
        (hl,) = struct.unpack_from("<I", buf, i)
        hdr = _fields(buf[i + 4:i + 4 + hl])
        (dl,) = struct.unpack_from("<I", buf, i + 4 + hl)
        data = buf[i + 8 + hl:i + 8 + hl + dl]
        i += 8 + hl + dl
        yield hdr, data


def read_messages(path):
    """-> (connections {id: (topic, type)}, messages [(conn id, time ns, bytes)]) in file order"""
    buf = open(path, "rb").read()
    magic = b"#ROSBAG V2.0\n"
    if not buf.startswith(magic):
        raise ValueError("not a rosbag v2.0 file")
    conns, msgs = {}, []

    def walk(b, start=0):
        for hdr, data in _records(b, start):
            op = hdr["op"][0]
            if op == OP_CHUNK:
                comp = hdr.get("compression", b"none")
                if comp == b"none":
                    walk(data)
                elif comp == b"bz2":
                    walk(bz2.decompress(data))
                else:
                    raise ValueError(f"unsupported chunk compression {comp!r}")
            elif op == OP_CONNECTION:
                cid = struct.unpack("<I", hdr["conn"])[0]
                ch = _fields(data)
                conns[cid] = (hdr["topic"].decode(), ch.get("type", b"").decode())
            elif op == OP_MSG:
                cid = struct.unpack("<I", hdr["conn"])[0]
                secs, nsecs = struct.unpack("<II", hdr["time"])
                msgs.append((cid, secs * 1_000_000_000 + nsecs, data))

    walk(buf, len(magic))
    return conns, msgs


def _header(b, i):
    seq, secs, nsecs, n = struct.unpack_from("<IIII", b, i)
    frame = b[i + 16:i + 16 + n].decode()
    return {"seq": seq, "stamp": (secs, nsecs), "frame_id": frame}, i + 16 + n


def parse_voxel_map(b):
    """planning_ros_msgs/VoxelMap -> dict(header, resolution f32, origin (3), dim (3 ints), data int8 x-fastest)"""
    hdr, i = _header(b, 0)
    (res,) = struct.unpack_from("<f", b, i)
    ox, oy, oz, dx, dy, dz = struct.unpack_from("<6d", b, i + 4)
    (n,) = struct.unpack_from("<I", b, i + 52)
    data = np.frombuffer(b, dtype=np.int8, count=n, offset=i + 56).copy()
    return {"header": hdr, "resolution": np.float32(res), "origin": (ox, oy, oz), "dim": (int(dx), int(dy), int(dz)), "data": data}


def parse_point_cloud(b):
    """sensor_msgs/PointCloud -> dict(header, points float64 (n,3) -- the Point32 values widened like cloud_to_vec,
    data_ros_utils.h:41-51)"""
    hdr, i = _header(b, 0)
    (n,) = struct.unpack_from("<I", b, i)
    pts = np.frombuffer(b, dtype="<f4", count=3 * n, offset=i + 4).reshape(n, 3).astype(np.float64)
    return {"header": hdr, "points": pts}


PARSERS = {"planning_ros_msgs/VoxelMap": parse_voxel_map, "sensor_msgs/PointCloud": parse_point_cloud}


def read_bag(path, topic):
    """All messages of `topic`, parsed (the reference takes `.back()`): read_bag<T>(file, topic, 0)"""
    conns, msgs = read_messages(path)
    out = []
    for cid, _, data in msgs:
        t, typ = conns.get(cid, (None, None))
        if t is not None and t.lstrip("/") == topic.lstrip("/"):
            if typ not in PARSERS:
                raise ValueError(f"no parser for message type {typ}")
            out.append(PARSERS[typ](data))
    return out
