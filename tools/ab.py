#!/usr/bin/env python3
"""A/B measurement of one library build (MPLX_LIB=build_tmp/libmplx_<variant>.so, tools/build_kernel_variant.sh) on the bench's
C4-ACC workload.  Prints one JSON line: kernel times per mode and a digest of every result word of every query, so that two
variants can be compared for speed AND for identical results.
  block : the 1024-query batch, cap 2 000 000, helpers auto (the driver's blocking step)
  bulk  : the same batch capped at 20 000 expansions per query, no helpers (256 compute units busy, no tail)
  tail  : query 1005 (the one that runs into the cap) alone, helpers auto
  c2    : BASELINE config 2 (256^3, single ACC query to the goal), helpers auto
usage: MPLX_LIB=... python tools/ab.py [reps] [modes ...]"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def word(r):
    return (r.status, r.traj_len, r.cost, r.n_expanded, r.n_nodes, r.n_edges, r.n_succ_finite, r.voxel_reads, r.n_push, r.expand_hash)


def digest(results):
    h = hashlib.sha256()
    for r in results:
        h.update(repr(word(r)).encode())
    return h.hexdigest()[:16]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    modes = sys.argv[2:] or ["block", "bulk", "tail"]
    from mpl_ros_amd import mapgen
    from tests import util
    out = {"lib": os.path.basename(os.environ.get("MPLX_LIB", "product"))}
    U = mapgen.control_lattice(1.0, 1, True)
    if any(m in modes for m in ("block", "bulk", "tail")):
        grid, origin, res, _, _, _ = mapgen.benchmark_map(512)
        grid = np.ascontiguousarray(grid)
        queries = mapgen.c4_queries(grid, origin, res, 1024, rank=0)
    for mode in modes:
        kw = dict(v_max=2.0, a_max=1.0, tol_pos=0.5)
        if mode in ("block", "bulk"):
            cap = 2_000_000 if mode == "block" else 20_000
            pools = mapgen.c4_pools(False, 1024, cap)
            mu, pl = util.make_gpu(grid, origin, res, U, n_slots=1024, max_expand=cap, max_nodes=pools["nodes"], max_edges=pools["edges"], max_log=pools["log"], **kw)
            if mode == "bulk":
                pl.setHelpers(0, 0)
            S = [util.gpu_wp(s) for s, g in queries]
            G = [util.gpu_wp(g) for s, g in queries]
            run = lambda: pl.planBatch(S, G)
        elif mode == "tail":
            s, g = queries[int(os.environ.get("QI", "1005"))]
            mu, pl = util.make_gpu(grid, origin, res, U, max_expand=2_000_000, max_nodes=1 << 24, max_edges=1 << 26, max_log=1 << 25, **kw)
            run = lambda: (pl.plan(util.gpu_wp(s), util.gpu_wp(g)), [pl.getResult()])[1]
        else:
            g2, o2, r2, s2, t2, _ = mapgen.benchmark_map(256)
            mu, pl = util.make_gpu(g2, o2, r2, U, max_nodes=1 << 22, max_edges=1 << 24, max_log=1 << 23, **kw)
            run = lambda: (pl.plan(util.gpu_wp(s2), util.gpu_wp(t2)), [pl.getResult()])[1]
        pl.setDeadline(100.0)
        if os.environ.get("AB_BUCKET_WIDTH"):  # (width of a coarse OPEN bucket in units of f; default 8 w dt)
            pl.setBucketWidth(float(os.environ["AB_BUCKET_WIDTH"]))
        if os.environ.get("AB_HELPERS") and mode != "bulk":  # "per_leader,reserved"
            a, b = os.environ["AB_HELPERS"].split(",")
            pl.setHelpers(int(a), int(b))
        ms, digs, n_exp = [], set(), 0
        for it in range(reps + 1):
            R = run()
            digs.add(digest(R))
            n_exp = sum(r.n_expanded for r in R)
            if it:
                ms.append(pl.lastKernelMs())
        out[mode] = {"kernel_ms": [round(x, 2) for x in ms], "min_ms": round(min(ms), 2), "mean_ms": round(sum(ms) / len(ms), 2), "expansions": n_exp,
                     "Mexp_per_s": round(n_exp / min(ms) / 1e3, 2), "digests": sorted(digs), "helpers": str(pl.helperStats())}
        print(json.dumps({mode: out[mode]}), file=sys.stderr, flush=True)
        del mu, pl
    print(json.dumps(out))


if __name__ == "__main__":
    main()
