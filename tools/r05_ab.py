#!/usr/bin/env python3
"""Round 5: what do the round-4 safeguards cost on the blocking C4-ACC step, now that the race underneath them is fixed?
One context, the bench's batch; per configuration of MPLX_X_FLAGS (read per launch): one warm-up batch, then `reps` timed ones; every
result of every batch is compared with the first configuration's.
  16 look-ahead rows behind an agent-scope release fence instead of the check word     32 TBL_DEAD_ID ahead of the parallel commit
  1024 rows unchecked (round 3)      2048 no wait for an earlier batch's claim (round 3)      4096 no TBL_DEAD_ID (round 3; needs 2048)
usage: r05_ab.py [reps] [flags ...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    flags = [int(x) for x in sys.argv[2:]] or [0, 16, 32, 48, 1024, 6144, 7168, 0]
    from mpl_ros_amd import mapgen
    from tests import util
    grid, origin, res, _, _, _ = mapgen.benchmark_map(512)
    grid = np.ascontiguousarray(grid)
    nq, cap = 1024, 2_000_000
    U = mapgen.control_lattice(1.0, 1, True)
    kw = dict(v_max=2.0, a_max=1.0, tol_pos=0.5, max_expand=cap)
    queries = mapgen.c4_queries(grid, origin, res, nq, rank=0)
    pools = mapgen.c4_pools(False, nq, cap)
    mu, pl = util.make_gpu(grid, origin, res, U, n_slots=1024, max_nodes=pools["nodes"], max_edges=pools["edges"], max_log=pools["log"], **kw)
    pl.setDeadline(60.0)
    S = [util.gpu_wp(s) for s, g in queries]
    G = [util.gpu_wp(g) for s, g in queries]
    word = lambda r: (r.status, r.traj_len, r.cost, r.n_expanded, r.n_nodes, r.n_edges, r.n_succ_finite, r.voxel_reads, r.n_push, r.expand_hash)
    ref, out = None, []
    for f in flags:
        os.environ["MPLX_X_FLAGS"] = str(f)
        ms, bad = [], 0
        for it in range(reps + 1):
            R = pl.planBatch(S, G)
            w = [word(r) for r in R]
            if ref is None:
                ref = w
            bad += sum(1 for a, b in zip(w, ref) if a != b)
            if it:
                ms.append(pl.lastKernelMs())
        rec = {"xflags": f, "kernel_ms": [round(x, 1) for x in ms], "min_ms": round(min(ms), 1), "mean_ms": round(sum(ms) / len(ms), 1), "differing_queries": bad}
        out.append(rec)
        print(json.dumps(rec), file=sys.stderr, flush=True)
    print(json.dumps({"kernel": pl.kernelName(), "runs": out}))


if __name__ == "__main__":
    main()
