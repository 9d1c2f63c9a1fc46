"""Regenerates tests/golden/plans.json: results of the CPU oracle on fixed scenarios.

These are regression pins of the restatement (and the HIP path is compared against them on the GPU
box, where the oracle is also re-run) -- they are NOT outputs of the reference, which cannot be
built here (oracle/mpl_oracle.h).  Run: python tests/golden/make_plan_fixtures.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mpl_ros_amd import mapgen  # noqa: E402
from oracle import orc  # noqa: E402
from tests import util  # noqa: E402


def scenario(name, grid, origin, res, control, num, start, goal, **kw):
    U = mapgen.control_lattice(1.0, num, True)
    P = util.make_oracle(grid, origin, res, control, U, **kw)
    st = P.plan(orc.waypoint(start[0], vel=start[1], control=control), orc.waypoint(goal, control=control))
    ids, _ = P.expanded()
    tr = P.traj()
    return {"name": name, "status": st, "cost": P.traj_cost if st == 0 else None, "n_expanded": len(ids),
            "expand_hash": str(util.expand_hash(ids)), "n_nodes": P.num_nodes(), "n_closed": P.num_closed(),
            "actions": tr["actions"].tolist(), "node_ids": tr["node_ids"].tolist(),
            "waypoints": [orc.wp_state(w, control).tolist() for w in tr["wps"]], "counters": P.counters()}


def main():
    out = []
    d = np.load(os.path.join(ROOT, "tests", "golden", "skir_map.npz"))
    out.append(scenario("skir_launch_query", d["grid"], d["origin"], float(d["res"]), orc.ACC, 1,
                        ((5.5, 5.5, 0.5), (1, 0, 0)), (1.5, 1.5, 5.5), v_max=2.0, a_max=1.0, tol_pos=0.5))
    for seed in (1, 2):
        grid, origin, res = util.small_map(96, seed=seed, occupancy=0.10)
        mapgen.carve_bubble(grid, (1.05, 1.05, 1.05), origin, res, 3)
        mapgen.carve_bubble(grid, (8.55, 8.55, 8.55), origin, res, 3)
        out.append(scenario(f"acc96_seed{seed}", grid, origin, res, orc.ACC, 1, ((1.05, 1.05, 1.05), (0, 0, 0)),
                            (8.55, 8.55, 8.55), v_max=2.0, a_max=1.0, tol_pos=0.5))
    grid, origin, res = util.small_map(96, seed=4, occupancy=0.10)
    mapgen.carve_bubble(grid, (1.05, 1.05, 1.05), origin, res, 3)
    out.append(scenario("jrk96_capped3000", grid, origin, res, orc.JRK, 2, ((1.05, 1.05, 1.05), (0, 0, 0)),
                        (8.55, 8.55, 8.55), v_max=2.0, a_max=1.0, j_max=1.0, tol_pos=0.5, max_expand=3000))
    with open(os.path.join(ROOT, "tests", "golden", "plans.json"), "w") as f:
        json.dump(out, f, indent=1)
    for s in out:
        print(s["name"], s["status"], s["cost"], s["n_expanded"], s["actions"])


if __name__ == "__main__":
    main()
