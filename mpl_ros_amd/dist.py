"""Multi-GPU execution of the query-sharded path (BASELINE.json config 4).

One process per GPU.  The path shards across QUERIES (one query is a serial A* pop chain and never
spans GPUs): rank 0 owns the voxel map, one broadcast (RCCL over xGMI with backend "nccl"; gloo in the
CPU tests) puts a replica into every rank's memory, each rank plans its share of the query stream with
no collective on the search path, and the per-query result rows are gathered on rank 0 at the end.

Everything here takes the `torch.distributed` module as an argument and is backend-agnostic, so the
world-size-2 gloo test (tests/test_multiproc_gloo.py) drives exactly the functions bench.py runs on
GPUs; only the per-rank `plan_fn` differs (the HIP planner there, a CPU checker in the test).
torch.distributed is transport only.
"""
import numpy as np

ROW = 8  # float64 per query row: query index, status, n_expanded, n_nodes, cost, hash_hi, hash_lo, traj_len


def shard_round_robin(n_items, rank, world):
    """Indices of the queries rank `rank` owns: q mod world == rank."""
    return list(range(rank, n_items, world))


def partition(queries, world, mode="lpt"):
    """Split the query stream over `world` ranks; returns one index list per rank.

    "rr"  : q mod world (BASELINE config 4 as written).
    "lpt" : expansions per query vary by more than 100x, so the stream is first ordered by the only
            predictor available before planning -- straight-line start-goal distance, longest first --
            and then dealt in a snake (0..w-1, w-1..0, ...): every rank gets the same mix of long and
            short queries, and each rank's own launch order stays longest-first."""
    n = len(queries)
    if mode == "rr":
        return [shard_round_robin(n, r, world) for r in range(world)]
    d = [-sum((s[i] - g[i]) ** 2 for i in range(3)) for s, g in queries]
    order = sorted(range(n), key=lambda i: (d[i], i))
    parts = [[] for _ in range(world)]
    for k, qi in enumerate(order):
        lap, pos = divmod(k, world)
        parts[pos if lap % 2 == 0 else world - 1 - pos].append(qi)
    return parts


def broadcast_map(dist, map_tensor, meta, src=0):
    """Broadcast the int8 voxel grid (flat tensor, already allocated on every rank) and its
    geometry header meta = [dx, dy, dz, ox, oy, oz, res] (float64 tensor of 7)."""
    dist.broadcast(meta, src=src)
    dist.broadcast(map_tensor, src=src)
    return map_tensor, meta


def result_row(qi, status, n_expanded, n_nodes, cost, expand_hash, traj_len):
    """Pack one query's result into a float64 row (all integers involved are < 2^53; the 64-bit hash is
    split into two 32-bit halves)."""
    return [float(qi), float(status), float(n_expanded), float(n_nodes), float(cost),
            float(int(expand_hash) >> 32), float(int(expand_hash) & 0xFFFFFFFF), float(traj_len)]


def gather_rows(dist, torch, rows, n_max, device="cpu"):
    """all_gather per-rank result rows (lists of ROW floats; at most n_max per rank, padded with -1) and
    return them as one list over all ranks (padding removed)."""
    world = dist.get_world_size()
    t = torch.full((n_max, ROW), -1.0, dtype=torch.float64, device=device)
    if rows:
        t[:len(rows)] = torch.tensor(rows, dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    merged = []
    for o in out:
        a = o.cpu().numpy()
        merged += [r.tolist() for r in a if r[0] >= 0]
    return merged


def merge_rows(n_items, rows):
    """Rows of all ranks -> list indexed by query (every query exactly once)."""
    out = [None] * n_items
    for r in rows:
        qi = int(r[0])
        assert out[qi] is None, f"query {qi} planned twice"
        out[qi] = r
    assert all(o is not None for o in out), "a query was not planned by any rank"
    return out


def run_sharded(dist, torch, rank, world, queries, plan_fn, mode="lpt", device="cpu", sync=None):
    """Plan `queries` (the same list on every rank) sharded over the ranks.

    plan_fn(indices) -> list of result_row(...) for those queries, in any order.
    sync()           -> drains the local device (torch.cuda.synchronize on GPUs); optional.
    Returns (merged rows indexed by query on every rank, seconds of the slowest rank for the plan step,
    per-rank [seconds, expansions]).  The timed region is barrier -> plan_fn -> barrier."""
    import time
    parts = partition(queries, world, mode)
    mine = parts[rank]
    if sync:
        sync()
    dist.barrier()
    t0 = time.perf_counter()
    rows = plan_fn(mine)
    if sync:
        sync()
    t_local = time.perf_counter() - t0
    dist.barrier()
    n_exp_local = float(sum(r[2] for r in rows))
    stat = torch.tensor([t_local, n_exp_local], dtype=torch.float64, device=device)
    stats = [torch.empty_like(stat) for _ in range(world)]
    dist.all_gather(stats, stat)
    per_rank = [s.cpu().tolist() for s in stats]
    merged = merge_rows(len(queries), gather_rows(dist, torch, rows, max(len(p) for p in parts), device))
    return merged, max(p[0] for p in per_rank), per_rank


def merge_sharded(n_items, world, per_rank_rows):
    """Inverse of shard_round_robin: per_rank_rows[r][i] belongs to query r + i*world."""
    out = [None] * n_items
    for r in range(world):
        for i, row in enumerate(per_rank_rows[r]):
            out[r + i * world] = row
    return out
