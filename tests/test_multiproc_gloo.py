"""CPU, world sizes 2 / 4 / 8, gloo: the query-sharded multi-process path (map broadcast, static round-robin
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
    queries = mapgen.random_queries(grid, origin, res, 11, mapgen.SplitMix64(99), min_dist=2.0)
    P = orc.Planner()
    P.set_map(grid, origin, res)
    P.set_config(orc.ACC, mapgen.control_lattice(), v_max=2.0, a_max=1.0)

    def plan_fn(indices):  # the per-rank search: the CPU oracle here, the HIP planner in bench.py
        rows = []
        for qi in indices:
            s, g = queries[qi]
            st = P.plan(orc.waypoint(s), orc.waypoint(g))
            ids, _ = P.expanded()
            h = 0
            for i in ids:
                h = (h * 0x100000001B3 + (int(i) + 1)) & ((1 << 64) - 1)
            rows.append(mdist.result_row(qi, st, len(ids), P.num_nodes(), P.traj_cost, h, P.traj()["n"] if st == 0 else 0))
        return rows

    out = {}
    for mode in ("lpt", "rr"):
        merged, t_max, per_rank = mdist.run_sharded(dist, torch, rank, world, queries, plan_fn, mode=mode)
        assert len(per_rank) == world and t_max >= max(p[0] for p in per_rank) - 1e-12
        assert sum(p[1] for p in per_rank) == sum(r[2] for r in merged)
        out[mode] = merged
    dist.barrier()
    if rank == 0:
        ref = sorted(plan_fn(range(len(queries))))  # single-process reference
        q.put((out["lpt"], out["rr"], ref, hash(map_t.numpy().tobytes())))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_query_sharding_matches_single_process(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    lpt, rr, ref, _ = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert lpt == ref and rr == ref  # every query planned exactly once, same rows as one process


def test_partitions_cover_every_query_once():
    from mpl_ros_amd import dist as mdist
    from mpl_ros_amd import mapgen
    rng = mapgen.SplitMix64(3)
    queries = [((rng.uniform(), rng.uniform(), rng.uniform()), (50 * rng.uniform(), 50 * rng.uniform(), 50 * rng.uniform())) for _ in range(1024)]
    for w in (1, 2, 4, 8):
        for mode in ("lpt", "rr"):
            parts = mdist.partition(queries, w, mode)
            assert sorted(sum(parts, [])) == list(range(1024)) and max(map(len, parts)) - min(map(len, parts)) <= 1
        d = [sum((s[i] - g[i]) ** 2 for i in range(3)) for s, g in queries]
        tot = [sum(d[i] for i in p) for p in mdist.partition(queries, w, "lpt")]
        assert max(tot) / min(tot) < 1.02  # the snake evens out the predictor
        for p in mdist.partition(queries, w, "lpt"):
            assert all(d[p[i]] >= d[p[i + 1]] for i in range(len(p) - 1))  # each rank launches longest-first


def test_shard_and_merge_are_inverse():
    from mpl_ros_amd import dist as mdist
    for n, w in ((10, 2), (1024, 8), (7, 4)):
        shards = [mdist.shard_round_robin(n, r, w) for r in range(w)]
        assert sorted(sum(shards, [])) == list(range(n))
        assert mdist.merge_sharded(n, w, shards) == list(range(n))
