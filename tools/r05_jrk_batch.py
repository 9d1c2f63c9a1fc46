#!/usr/bin/env python3
"""Round 5, first GPU call: the C4-JRK 1024-query batch of tests/test_gpu_scale.py (the test the driver's round-4 run stalled in)
under the launch guard: a launch that does not end is aborted at the deadline and the watch records say where it sat.
usage: r05_jrk_batch.py [repeats] [deadline seconds] [n_slots] [helpers per leader (-1 auto, 0 off)]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    deadline = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
    n_slots = int(sys.argv[3]) if len(sys.argv) > 3 else 768
    helpers = int(sys.argv[4]) if len(sys.argv) > 4 else -1
    from mpl_ros_amd import mapgen
    from mpl_ros_amd._capi import MplxError
    from oracle import orc
    from tests import util
    grid, origin, res, _, _, _ = mapgen.benchmark_map(512)
    grid = np.ascontiguousarray(grid)
    nq, cap = 1024, 20000
    U = mapgen.control_lattice(1.0, 2, True)
    kw = dict(v_max=2.0, a_max=1.0, j_max=1.0, tol_pos=0.5, max_expand=cap)
    queries = mapgen.c4_queries(grid, origin, res, nq, rank=0)
    pools = mapgen.c4_pools(True, nq, cap)
    mu, pl = util.make_gpu(grid, origin, res, U, n_slots=n_slots, max_nodes=pools["nodes"], max_edges=pools["edges"], max_log=pools["log"], **kw)
    pl.setHelpers(helpers, -1)
    pl.setDeadline(deadline)
    S = [util.gpu_wp(s, control=orc.JRK) for s, g in queries]
    G = [util.gpu_wp(g, control=orc.JRK) for s, g in queries]
    word = lambda r: (r.status, r.traj_len, r.cost, r.n_expanded, r.n_nodes, r.n_edges, r.n_succ_finite, r.voxel_reads, r.n_push, r.expand_hash)
    out = {"n_slots": n_slots, "helpers": helpers, "deadline_s": deadline, "runs": []}
    ref = None
    for it in range(reps):
        t0 = time.time()
        try:
            R = pl.planBatch(S, G)
            words = [word(r) for r in R]
            if ref is None:
                ref = words
            out["runs"].append({"ok": True, "wall_s": round(time.time() - t0, 3), "kernel_ms": round(pl.lastKernelMs(), 1),
                                "expansions": int(sum(r.n_expanded for r in R)), "status_hist": np.bincount([r.status for r in R]).tolist(),
                                "differs_from_first": sum(1 for a, b in zip(words, ref) if a != b), "helper_stats": pl.helperStats()})
        except MplxError as e:
            out["runs"].append({"ok": False, "wall_s": round(time.time() - t0, 3), "error": str(e)})
        print(json.dumps(out["runs"][-1]), file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
