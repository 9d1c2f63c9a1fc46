#!/usr/bin/env python3
"""Diagnostic (round 4): does the BLOCKING C4-ACC batch still reproduce bit for bit when something else keeps the device busy?

The streamed leg of bench.py showed a few queries per 20 batches that differ from the blocking step -- states created twice -- only
when two lanes' launches overlap.  Overlap means two things at once: (a) workgroups of a launch start late, on compute units another
launch has just left, and (b) the other lane's hipMemset kernels and searches run on the same compute units and memory system.  This
probe keeps (b) and drops (a): ONE context, mplx_plan_batch as usual, while a background thread fills a large buffer over and over on
a side stream.  usage: fill_load_probe.py [batches] [mode] [lattice]   mode: fill (default) | read | alu | none; lattice: acc (default:
the C4-ACC batch, <32,16,ACC,help>) | jrk (the C4-JRK batch, 125 inputs, cap 20 000, <128,4,JRK,help>)

Round 5: the process runs torch FIRST (its bundled HIP runtime must initialise before libmplx's: the other way round torch finds no
device), which is why tests/test_zz_jitter.py runs this file as a subprocess at the end of a session that has long been planning.
"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(n_batches=8, mode="fill", lattice="acc"):
    import torch
    torch.cuda.init()
    from mpl_ros_amd import mapgen
    from mpl_ros_amd.planner import ACC, JRK, VoxelMapPlanner, VoxelMapUtil, Waypoint3D
    jrk = lattice == "jrk"
    control, cap = (JRK, 20000) if jrk else (ACC, 2_000_000)
    dev = torch.device("cuda", 0)
    n, res, origin = 512, 0.1, (0.0, 0.0, 0.0)
    grid, _, _, _, _, _ = mapgen.benchmark_map(n)
    map_t = torch.from_numpy(grid.reshape(-1)).to(dev)
    mu = VoxelMapUtil(0)
    mu.setMapDevice(map_t.data_ptr(), origin, (n, n, n), res)
    queries = mapgen.c4_queries(grid, origin, res, 1024, rank=0)
    caps = mapgen.c4_pools(jrk, 1024, cap)
    pl = VoxelMapPlanner(False)
    pl.setMapUtil(mu)
    pl.setVmax(2.0); pl.setAmax(1.0); pl.setDt(1.0)
    if jrk:
        pl.setJmax(1.0)
    pl.setU(mapgen.control_lattice(1.0, 2 if jrk else 1, True))
    pl.setTol(0.5); pl.setMaxNum(cap)
    pl.setCapacity(768 if jrk else 1024, caps["nodes"], caps["edges"], caps["log"])
    pl.setHelpers(-1, -1)

    def wp(p):
        w = Waypoint3D(control)
        w.pos = np.array(p, dtype=np.float64)
        return w
    starts = [wp(q[0]) for q in queries]
    goals = [wp(q[1]) for q in queries]
    key = lambda r: (r.status, r.traj_len, r.cost, r.n_expanded, r.n_nodes, r.n_edges, r.n_succ_finite, r.voxel_reads, r.expand_hash)
    want = [key(r) for r in pl.planBatch(starts, goals)]
    again = [key(r) for r in pl.planBatch(starts, goals)]
    quiet_ms = pl.lastKernelMs()
    assert again == want, "the quiet blocking batch does not repeat"
    recycle = os.environ.get("FILL_NO_RECYCLE") != "1"
    if recycle:  # (round 6) the batches under load run like bench.py's: recycled pools of about half the size, compared with the un-recycled quiet batch
        small = mapgen.c4_pools(jrk, 1024, cap, per_q=(0 if jrk else 250_000))
        if jrk:
            small = {k: v // 2 for k, v in small.items()}
        pl.setPoolRecycling(True)
        pl.setCapacity(768 if jrk else 1024, small["nodes"], small["edges"], small["log"])
    stop = threading.Event()
    fills = [0]

    def background():
        torch.cuda.set_device(0)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            if mode == "fill":    # write traffic: the other lane's hipMemset of its state table and look-ahead cache
                buf = torch.empty(16 << 30, dtype=torch.uint8, device=dev)
            elif mode == "read":  # read traffic only: HBM contention and L2 churn, nothing written
                buf = torch.zeros(4 << 30, dtype=torch.float32, device=dev)
            else:                 # "alu": waves on every SIMD, a working set that stays in the L2 (1 MB): no memory-system load
                buf = torch.rand(1 << 18, dtype=torch.float32, device=dev)
            while not stop.is_set():
                if mode == "fill":
                    buf.fill_(255)
                    buf[: 7 << 30].zero_()
                elif mode == "read":
                    buf.sum()
                else:
                    for _ in range(2000):
                        buf = torch.sin(buf) * 1.0001 + 0.1
                side.synchronize()
                fills[0] += 1
                time.sleep(0.2)  # (a lane submits every ~0.65 s in the streamed leg: bursts, not a continuous load)
    th = None
    if mode != "none":
        th = threading.Thread(target=background, daemon=True)
        th.start()
        time.sleep(1.0)
    bad = []
    ms = []
    for b in range(n_batches):
        got = [key(r) for r in pl.planBatch(starts, goals)]
        ms.append(pl.lastKernelMs())
        for qi, (g, w) in enumerate(zip(got, want)):
            if g != w:
                bad.append({"batch": b, "query": qi, "status": int(g[0]), "d_expanded": int(g[3] - w[3]), "d_nodes": int(g[4] - w[4])})
    stop.set()
    if th:
        th.join(timeout=30)
    return ({"probe": "blocking batch with a background fill load", "mode": mode, "kernel": pl.kernelName(), "batches": n_batches, "fill_rounds": fills[0],
                      "quiet_kernel_ms": quiet_ms, "kernel_ms": ms, "pool_recycling_under_load": recycle, "mismatching_queries": len(bad), "detail": bad[:12]})


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 8, sys.argv[2] if len(sys.argv) > 2 else "fill", sys.argv[3] if len(sys.argv) > 3 else "acc")))
