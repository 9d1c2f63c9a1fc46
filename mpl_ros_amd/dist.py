"""Multi-GPU plumbing for the query-sharded path: one process per GPU, one map replica
per GPU distributed by a broadcast (RCCL over xGMI with backend "nccl", gloo in the CPU tests),
queries partitioned statically, no collective on the search path, results gathered at the end.
torch.distributed is used for the transport only.
"""
import numpy as np


def shard_round_robin(n_items, rank, world):
    """Indices of the queries rank `rank` owns: q mod world == rank."""
    return list(range(rank, n_items, world))


def broadcast_map(dist, map_tensor, meta, src=0):
    """Broadcast the int8 voxel grid (flat tensor, already allocated on every rank) and its
    geometry header meta = [dx, dy, dz, ox, oy, oz, res] (float64 tensor of 7)."""
    dist.broadcast(meta, src=src)
    dist.broadcast(map_tensor, src=src)
    return map_tensor, meta


def gather_int64(dist, torch, values, device="cpu"):
    """all_gather a small list of int64 per rank -> array (world, len(values))."""
    world = dist.get_world_size()
    t = torch.tensor(values, dtype=torch.int64, device=device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return np.stack([o.cpu().numpy() for o in out])


def merge_sharded(n_items, world, per_rank_rows):
    """Inverse of shard_round_robin: per_rank_rows[r][i] belongs to query r + i*world."""
    out = [None] * n_items
    for r in range(world):
        for i, row in enumerate(per_rank_rows[r]):
            out[r + i * world] = row
    return out
