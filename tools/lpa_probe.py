"""LPA* vs fresh A* on the replanner scenario of tests/test_lpa.py (simple map, add_cloud / clear_cloud / subtree):
expansions and kernel time of each plan.  usage: python tools/lpa_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util
from tests.test_lpa import Scenario, gpu_pair, set_gpu_map, START, START_V, GOAL
sc = Scenario()
mu, a, l = gpu_pair(sc)
s, g = util.gpu_wp(START, vel=START_V), util.gpu_wp(GOAL)


def both(tag, s):
    a.plan(s, g); ra = a.getResult(); ma = a.lastKernelMs()
    l.plan(s, g); rl = l.getResult(); ml = l.lastKernelMs()
    print(f"{tag}: fresh A* {ra.n_expanded} expansions {ma:.3f} ms (cost {ra.cost}) | LPA* {rl.n_expanded} expansions {ml:.3f} ms (cost {rl.cost})", flush=True)


both("first plan", s)
new_obs = sc.add((12.55, 9.55, 0.025), (12.55, 11.05, 0.025))
set_gpu_map(mu, sc)
print("updateBlockedNodes:", l.updateBlockedNodes(new_obs), "entries changed,", len(new_obs), "cells")
both("after add_cloud", s)
cl = sc.clear((12.75, 9.55, 0.025), (12.65, 11.95, 0.025))
set_gpu_map(mu, sc)
print("updateClearedNodes:", l.updateClearedNodes(cl), "entries changed,", len(cl), "cells")
both("after clear_cloud", s)
tg = l.getTraj()
l.getSubStateSpace(1)
w1 = tg.getWaypoints()[1]
both("after getSubStateSpace(1)", util.gpu_wp(tuple(w1.pos), vel=tuple(w1.vel)))
