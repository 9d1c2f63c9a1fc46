"""One CPU-baseline worker PROCESS (test infrastructure, see oracle/mpl_oracle.h): plans the queries it is fed
on stdin with the CPU restatement and answers one JSON line each.  bench.py's cpu_baseline leg starts N of
these (one per core): separate address spaces, so the planners do not serialise on one process's page-fault /
allocator locks the way N threads of one process do; the voxel map is ONE read-only file in /dev/shm mapped by
all of them (orc_set_map_shared).

    python oracle/cpu_worker.py '<json config>'      config: map (npy path), origin, res, control, U, kw, native
    stdin : "<index> sx sy sz gx gy gz [cap]\\n" ...  stdout: {"i":..,"status":..,"n_expanded":..,...}\\n
    cap (optional): expansion cap of THIS query (a long query of the sample is timed over its first `cap` expansions)
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    cfg = json.loads(sys.argv[1])
    if cfg.get("cpu") is not None:  # one worker per physical core, spread over the sockets
        try:
            os.sched_setaffinity(0, {int(cfg["cpu"])})
        except OSError:
            pass
    from oracle import orc
    if cfg.get("native"):
        orc.use_native()
    grid = np.load(cfg["map"], mmap_mode="r")
    P = orc.Planner()
    P.set_map_shared(grid, cfg["origin"], cfg["res"])
    control = cfg["control"]
    P.set_config(control, np.array(cfg["U"], dtype=np.float64), **cfg["kw"])
    print(json.dumps({"ready": True}), flush=True)
    capped = False
    for line in sys.stdin:
        f = line.split()
        if not f:
            break
        i = int(f[0])
        s, g = [float(x) for x in f[1:4]], [float(x) for x in f[4:7]]
        cap = int(f[7]) if len(f) > 7 else None
        if cap is not None or capped:
            P.set_config(control, np.array(cfg["U"], dtype=np.float64), **(dict(cfg["kw"], max_expand=cap) if cap is not None else cfg["kw"]))
            capped = cap is not None
        P.reset_counters()
        t0 = time.perf_counter()
        st = P.plan(orc.waypoint(s, control=control), orc.waypoint(g, control=control))
        dt = time.perf_counter() - t0
        out = {"i": i, "status": st, "n_expanded": P.counters()["n_expansions"], "n_nodes": P.num_nodes(), "cost": P.traj_cost,
               "hash": P.expand_hash(), "seconds": dt, "actions": P.traj()["actions"].tolist() if st == orc.OK else None}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
