import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from mpl_ros_amd import poly_map as pm
worlds, starts, goals = pm.team2_tick()
team = pm.PolyTeam(); team.configure(pm.ACC, pm.U9, dt=0.5, v_max=2.0, a_max=1.0, w=10.0); team.set_worlds(worlds); team.set_capacity(16, 1<<21, 1<<23, 1<<22)
for it in range(2):
    R = team.plan_batch(np.arange(16), starts, goals, max_expand=20000)
k = int(np.argmax([r.n_expanded for r in R])); r = R[k]
print("longest robot", k, "expansions", r.n_expanded, "nodes", r.n_nodes, "push", r.n_push, "refill", r.n_refill, "evict", r.n_evict, "kernel ms", team.last_kernel_ms(), team.cycles(k))
