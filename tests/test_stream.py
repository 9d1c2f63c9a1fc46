"""Streamed query batches (include/mplx.h: mplx_plan_batch_submit / _wait, mplx_stream_*): several batches in flight on one map
replica.  north_star: "many independent start/goal queries (multi-robot, replanning) shard one-query-per-stream"; the
reference's own statement of the independence is robot_team.hpp:60-66 (every robot plans on its own).  What is checked: a
batch gives the same results -- every field, every query, the trajectories -- whether it runs alone and blocking, or
overlapped with other batches on other lanes, with or without a helper limit."""
import numpy as np
import pytest

from mpl_ros_amd import mapgen
from oracle import orc
from tests import util

KW = dict(v_max=2.0, a_max=1.0, tol_pos=0.5)


def _queries(grid, origin, res, n, seed):
    rng = mapgen.SplitMix64(seed)
    return mapgen.random_queries(grid, origin, res, n, rng, min_dist=3.0)


def _tuple(r):
    return (r.status, r.traj_len, r.cost, r.n_expanded, r.n_closed, r.n_nodes, r.n_edges, r.n_succ, r.n_succ_finite, r.voxel_reads, r.n_push, r.n_reopen,
            r.expand_hash)


@pytest.mark.gpu
def test_submit_wait_is_plan_batch_and_a_stream_overlaps_batches_without_changing_them():
    grid, origin, res = util.small_map(64, seed=11)
    U = mapgen.control_lattice(1.0, 1, True)
    mu, pl = util.make_gpu(grid, origin, res, U, n_slots=64, max_nodes=1 << 21, max_edges=1 << 23, max_log=1 << 22, max_expand=30000, **KW)
    sets = [_queries(grid, origin, res, 40, 100 + k) for k in range(3)]
    wps = [([util.gpu_wp(s) for s, _ in qs], [util.gpu_wp(g) for _, g in qs]) for qs in sets]
    # blocking reference runs (the path the parity suite checks against the CPU oracle); one query of each is re-checked here
    ref, ref_act = [], []
    for S, G in wps:
        R = pl.planBatch(S, G)
        ref.append([_tuple(r) for r in R])
        ref_act.append([pl.getTraj(q).actions.copy() for q in range(len(S))])
    P = util.make_oracle(grid, origin, res, orc.ACC, U, max_expand=30000, **KW)
    s0, g0 = sets[0][0]
    st = P.plan(orc.waypoint(s0), orc.waypoint(g0))
    assert st == ref[0][0][0] and util.expand_hash(P.expanded()[0]) == ref[0][0][-1]
    # the asynchronous pair on the planner's own context
    pl.planBatchSubmit(*wps[1])
    R = pl.planBatchWait()
    assert [_tuple(r) for r in R] == ref[1]
    assert all(np.array_equal(pl.getTraj(q).actions, ref_act[1][q]) for q in range(len(R)))
    # a second submit before the wait is refused, and leaves the outstanding batch intact
    pl.planBatchSubmit(*wps[2])
    with pytest.raises(Exception):
        pl.planBatchSubmit(*wps[0])
    assert [_tuple(r) for r in pl.planBatchWait()] == ref[2]
    # a stream of two lanes: twelve batches, two in flight at any time, with and without a helper limit
    for limit in (-1, 2):
        st = pl.stream(2)
        st.configure(64, 1 << 21, 1 << 23, 1 << 22, -1, 0, 0, limit)
        order = [k % 3 for k in range(12)]
        inflight, seen = [], 0
        for k in order:
            if len(inflight) == 2:
                t, kk = inflight.pop(0)
                R = st.wait(t)
                assert [_tuple(r) for r in R] == ref[kk], (limit, seen)
                for q in (0, 7, 39):
                    assert np.array_equal(st.trajActions(q, R[q].traj_len), ref_act[kk][q])
                seen += 1
            inflight.append((st.submit(*wps[k]), k))
        with pytest.raises(Exception):  # every lane busy
            st.submit(*wps[0])
        for t, kk in inflight:
            assert [_tuple(r) for r in st.wait(t)] == ref[kk]
            seen += 1
        assert seen == 12
        st.close()
    # the lanes follow the parent's map: edit it (dilate) between batches and the next submit plans on the edited map; with a
    # batch in flight the submit that would have to re-adopt the map is refused
    st = pl.stream(2)
    st.configure(64, 1 << 21, 1 << 23, 1 << 22, -1, 0, 0, 4)
    assert [_tuple(r) for r in st.wait(st.submit(*wps[0]))] == ref[0]
    t_busy = st.submit(*wps[1])
    mu.dilate([(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0)])
    with pytest.raises(Exception):
        st.submit(*wps[2])
    st.wait(t_busy)
    dil = [_tuple(r) for r in st.wait(st.submit(*wps[0]))]
    assert dil == [_tuple(r) for r in pl.planBatch(*wps[0])] and dil != ref[0]
    st.close()
    dz, dy, dx = grid.shape
    mu.setMap(origin, (dx, dy, dz), grid.ravel(), res)
    mu.freeUnknown()
    # the planner's own context still plans (its pools were not touched by the lanes)
    assert [_tuple(r) for r in pl.planBatch(*wps[0])] == ref[0]
    pl.releasePools()
    assert [_tuple(r) for r in pl.planBatch(*wps[0])] == ref[0]


@pytest.mark.gpu
def test_stream_lanes_follow_the_parents_setup_and_refuse_an_auxiliary_map():
    """a PlanStream must plan with what planBatch would plan with.  Setters called on the planner after stream() reach
    the lanes with the next submit (epsilon, max_num here); a potential field -- which the lanes cannot carry -- makes the
    submit fail loudly instead of planning without it."""
    from mpl_ros_amd._capi import MplxError
    grid, origin, res = util.small_map(64, seed=5)
    U = mapgen.control_lattice(1.0, 1, True)
    mu, pl = util.make_gpu(grid, origin, res, U, n_slots=32, max_nodes=1 << 21, max_edges=1 << 23, max_log=1 << 22, max_expand=30000, **KW)
    qs = _queries(grid, origin, res, 24, 77)
    S, G = [util.gpu_wp(s) for s, _ in qs], [util.gpu_wp(g) for _, g in qs]
    ref = [_tuple(r) for r in pl.planBatch(S, G)]
    st = pl.stream(2)
    st.configure(32, 1 << 21, 1 << 23, 1 << 22, -1, 0, 0, -1)
    assert [_tuple(r) for r in st.wait(st.submit(S, G))] == ref
    pl.setEpsilon(2.0)
    pl.setMaxNum(4000)
    got = [_tuple(r) for r in st.wait(st.submit(S, G))]  # (the submit carries the new set-up to the idle lanes)
    ref2 = [_tuple(r) for r in pl.planBatch(S, G)]
    assert got == ref2 and ref2 != ref
    pl.setPotentialRadius((0.4, 0.4, 0.3)); pl.setPotentialWeight(3.0)
    pl.updatePotentialMap((2.0, 2.0, 2.0))
    with pytest.raises(MplxError) as e:
        st.submit(S, G)
    assert "auxiliary map" in str(e.value)
    st.close()
