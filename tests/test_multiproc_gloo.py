"""CPU, world_size 2, gloo: the query-sharded multi-process path (map broadcast, static round-robin
shard, no collective on the search path, gather at the end).  The per-rank search is done by the CPU
oracle here; on GPUs bench.py runs the same plumbing with backend "nccl" (= RCCL) and the HIP planner."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from mpl_ros_amd import dist as mdist
    from mpl_ros_amd import mapgen
    from oracle import orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 48
    meta = torch.zeros(7, dtype=torch.float64)
    if rank == 0:
        grid, _ = mapgen.random_box_map((n, n, n), seed=5, occupancy=0.08, edge=(2, 8))
        map_t = torch.from_numpy(grid.reshape(-1).copy())
        meta[:] = torch.tensor([n, n, n, 0.0, 0.0, 0.0, 0.1])
    else:
        map_t = torch.empty(n * n * n, dtype=torch.int8)
    mdist.broadcast_map(dist, map_t, meta, src=0)
    grid = map_t.numpy().reshape(n, n, n)
    origin, res = tuple(meta[3:6].tolist()), float(meta[6])
    queries = mapgen.random_queries(grid, origin, res, 10, mapgen.SplitMix64(99), min_dist=2.0)
    P = orc.Planner()
    P.set_map(grid, origin, res)
    P.set_config(orc.ACC, mapgen.control_lattice(), v_max=2.0, a_max=1.0)
    mine = mdist.shard_round_robin(len(queries), rank, world)
    rows = []
    for qi in mine:
        s, g = queries[qi]
        st = P.plan(orc.waypoint(s), orc.waypoint(g))
        rows.append([qi, st, P.num_closed(), int(round(P.traj_cost * 1000)) if st == 0 else -1])
    while len(rows) < (len(queries) + world - 1) // world:
        rows.append([-1, -1, -1, -1])
    allrows = mdist.gather_int64(dist, torch, rows)
    dist.barrier()
    if rank == 0:
        per_rank = [[r for r in allrows[k].tolist() if r[0] >= 0] for k in range(world)]
        merged = mdist.merge_sharded(len(queries), world, per_rank)
        # single-process reference
        ref = []
        for qi, (s, g) in enumerate(queries):
            st = P.plan(orc.waypoint(s), orc.waypoint(g))
            ref.append([qi, st, P.num_closed(), int(round(P.traj_cost * 1000)) if st == 0 else -1])
        q.put((merged, ref, hash(map_t.numpy().tobytes())))
    dist.destroy_process_group()


def test_two_rank_query_sharding_matches_single_process():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged, ref, _ = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert merged == ref


def test_shard_and_merge_are_inverse():
    from mpl_ros_amd import dist as mdist
    for n, w in ((10, 2), (1024, 8), (7, 4)):
        shards = [mdist.shard_round_robin(n, r, w) for r in range(w)]
        assert sorted(sum(shards, [])) == list(range(n))
        assert mdist.merge_sharded(n, w, shards) == list(range(n))
