#!/usr/bin/env python3
"""Round 5: is the C4-JRK 1024-query batch (tests/test_gpu_scale.py; <128,4,JRK,help>) repeatable?  Round 4's driver run stalled in it;
round 5's first GPU call found 34-37 of 1024 queries differing from run to run on a quiet device.  The batch is planned under a list of
configurations -- helpers off (the reference: no other workgroup touches a query's data), helpers on, diagnostic switches
(MPLX_X_FLAGS, read per launch) -- and every run is compared with the first helper-less run, field by field.
usage: r05_jrk_batch.py [deadline seconds] [n_slots] [config ...]     config = helpers:xflags:repeats, e.g. 0:0:2 -1:0:3 -1:2:3"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FIELDS = ("status", "traj_len", "cost", "n_expanded", "n_nodes", "n_edges", "n_succ_finite", "voxel_reads", "n_push", "expand_hash")


def main():
    deadline = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    n_slots = int(sys.argv[2]) if len(sys.argv) > 2 else 768
    configs = [tuple(int(x) for x in a.split(":")) for a in sys.argv[3:]] or [(0, 0, 2), (-1, 0, 3)]
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
    pl.setDeadline(deadline)
    S = [util.gpu_wp(s, control=orc.JRK) for s, g in queries]
    G = [util.gpu_wp(g, control=orc.JRK) for s, g in queries]
    word = lambda r: tuple(getattr(r, f) for f in FIELDS)
    import ctypes as C

    def audit(q):
        """raw node records of query q of the last batch: states against keys, duplicate keys"""
        ctx = pl._ctx()
        n, rs = C.c_uint64(0), C.c_int32(0)
        ctx.lib.mplx_debug_query_records(ctx.h, q, 0, None, C.byref(n), C.byref(rs))
        buf = np.empty(int(n.value) * int(rs.value), dtype=np.uint8)
        ctx.check(ctx.lib.mplx_debug_query_records(ctx.h, q, buf.nbytes, buf.ctypes.data, C.byref(n), C.byref(rs)))
        rec = buf.reshape(int(n.value), int(rs.value))
        keys = rec[:, 24:24 + 36].copy().view(np.int32)           # JRK: 9 key integers (pos, vel, acc per axis)
        st = rec[:, 64:64 + 80].copy().view(np.float64)           # pos3 vel3 acc3 t
        rnd = lambda x: (np.sign(x) * np.floor(np.abs(x) + 0.5)).astype(np.int64)
        want = np.empty((rec.shape[0], 9), dtype=np.int64)
        for ax in range(3):
            want[:, 3 * ax] = rnd(st[:, ax] / 0.01)
            want[:, 3 * ax + 1] = rnd(st[:, 3 + ax] / 0.1)
            want[:, 3 * ax + 2] = rnd(st[:, 6 + ax] / 0.1)
        badrow = np.nonzero((want != keys).any(axis=1))[0]
        order = np.lexsort(keys.T[::-1])
        sk = keys[order]
        same = (sk[1:] == sk[:-1]).all(axis=1)
        dups = [(int(min(order[i], order[i + 1])), int(max(order[i], order[i + 1]))) for i in np.nonzero(same)[0]]
        info = {"query": q, "n_records": int(n.value), "states_not_matching_their_key": int(len(badrow)), "duplicate_keys": len(dups)}
        if len(badrow):
            i = int(badrow[0])
            info["first_bad"] = {"id": i, "key": keys[i].tolist(), "key_of_state": want[i].tolist(), "state": [float(x) for x in st[i]]}
            info["bad_ids"] = [int(x) for x in badrow[:16]]
        if dups:
            info["dup_id_pairs"] = dups[:12]
            info["dup_gaps"] = sorted(b - a for a, b in dups)[:40]
            # runs of consecutive ids with one key: lengths, and the records around the first run
            ids = sorted(set(x for p_ in dups for x in p_))
            runs, cur = [], [ids[0]]
            for x in ids[1:]:
                if x == cur[-1] + 1 and (keys[x] == keys[cur[-1]]).all():
                    cur.append(x)
                else:
                    runs.append(cur); cur = [x]
            runs.append(cur)
            info["run_lengths"] = [len(r) for r in runs][:40]
            info["run_starts"] = [r[0] for r in runs][:40]
            r0 = runs[0]
            g = rec[:, 0:8].copy().view(np.float64)[:, 0]
            h = rec[:, 8:16].copy().view(np.float64)[:, 0]
            fl = rec[:, 16:20].copy().view(np.uint32)[:, 0]
            pr = rec[:, 20:24].copy().view(np.uint32)[:, 0]
            lo, hi = max(0, r0[0] - 3), min(rec.shape[0], r0[-1] + 4)
            info["around_first_run"] = [{"id": i, "key": keys[i].tolist(), "g": float(g[i]), "h": float(h[i]), "flags": int(fl[i]), "pred": int(pr[i]), "t": float(st[i, 9])}
                                        for i in range(lo, hi)][:24]
        return info
    out = {"n_slots": n_slots, "deadline_s": deadline, "runs": []}
    ref = None
    for helpers, xflags, reps in configs:
        pl.setHelpers(helpers, -1)
        os.environ["MPLX_X_FLAGS"] = str(xflags)
        for it in range(reps):
            t0 = time.time()
            rec = {"helpers": helpers, "xflags": xflags}
            try:
                R = pl.planBatch(S, G)
                words = [word(r) for r in R]
                if ref is None:
                    ref = words
                bad = [i for i, (a, b) in enumerate(zip(words, ref)) if a != b]
                by_field = {f: sum(1 for i in bad if words[i][k] != ref[i][k]) for k, f in enumerate(FIELDS)}
                if bad:
                    rec["audit"] = [audit(i) for i in bad[:3]]
                elif helpers == 0 and it == 0:
                    rec["audit"] = [audit(i) for i in (28, 142)]
                rec.update(ok=True, wall_s=round(time.time() - t0, 3), kernel_ms=round(pl.lastKernelMs(), 1), kernel=pl.kernelName(),
                           expansions=int(sum(r.n_expanded for r in R)), status_hist=np.bincount([r.status for r in R], minlength=8).tolist(),
                           differs_from_reference=len(bad), differing_fields=by_field,
                           examples=[{"query": i, "got": [str(x) for x in words[i]], "ref": [str(x) for x in ref[i]]} for i in bad[:4]])
            except MplxError as e:
                rec.update(ok=False, wall_s=round(time.time() - t0, 3), error=str(e))
            out["runs"].append(rec)
            print(json.dumps({k: v for k, v in rec.items() if k not in ("examples", "differing_fields")}), file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
