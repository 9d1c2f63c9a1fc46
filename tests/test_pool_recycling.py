"""Round 6: pool recycling and the epoch-tagged state table (no reference counterpart: the reference grows std containers).

* mplx_set_pool_recycling: a finished query of a batch hands its pool chunks back; the batch then runs in pools far smaller than the
  sum of its queries' state spaces -- with every result word, trajectory and counter what it is without recycling.
* the shared state table is not cleared between batches any more: its slots carry the launch epoch, which wraps after 255 launches
  (then, and only then, the table is cleared).  Hundreds of consecutive batches on one context must repeat exactly.
Both are checked against the un-recycled run of the same library, whose parity with the oracle is the business of tests/test_gpu_parity.py
and tests/test_gpu_scale.py."""
import ctypes as C

import numpy as np
import pytest

from mpl_ros_amd import _capi, mapgen
from tests import util

pytestmark = pytest.mark.gpu
KW = dict(v_max=2.0, a_max=1.0, tol_pos=0.5)


def word(r):
    return (r.status, r.traj_len, r.cost, r.n_expanded, r.n_nodes, r.n_edges, r.n_succ, r.n_succ_finite, r.voxel_reads, r.n_push, r.expand_hash)


def batch(n, nq, seed=5):
    grid, _ = mapgen.random_box_map((n, n, n), seed=seed, occupancy=0.10, edge=(3, 9))
    origin, res = (0.0, 0.0, 0.0), 0.1
    queries = mapgen.random_queries(grid, origin, res, nq, mapgen.SplitMix64(1234 + seed), min_dist=0.5 * n * res)
    return grid, origin, res, queries


@pytest.mark.parametrize("helpers", [0, -1])
def test_recycled_batch_equals_the_unrecycled_batch_in_much_smaller_pools(helpers):
    grid, origin, res, queries = batch(160, 96)
    U = mapgen.control_lattice(1.0, 1, True)
    S = [util.gpu_wp(s) for s, g in queries]
    G = [util.gpu_wp(g) for s, g in queries]
    SLOTS = 4  # (workgroups that lead at a time: what the recycled pools have to hold)
    mu, pl = util.make_gpu(grid, origin, res, U, n_slots=SLOTS, max_nodes=1 << 24, max_edges=1 << 26, max_log=1 << 25, max_expand=150000, **KW)
    pl.setHelpers(helpers, -1)
    ref = pl.planBatch(S, G)
    ref_traj = [pl.getTraj(k).actions.copy() for k in range(len(S))]
    total_nodes, total_edges = sum(r.n_nodes for r in ref), sum(r.n_edges for r in ref)
    biggest = max(r.n_nodes for r in ref)
    assert sum(r.status == 0 for r in ref) > 48 and total_nodes > 16 * 32768
    del pl
    # pools for SLOTS of the biggest query (rounded up to chunks): a fraction of the batch's total
    cap_n = SLOTS * (biggest + 2 * 32768)
    cap_e = SLOTS * (max(r.n_edges for r in ref) + 2 * 65536)
    assert cap_n < 0.6 * total_nodes and cap_e < 0.6 * total_edges, (cap_n, total_nodes, cap_e, total_edges)
    mu2, pr = util.make_gpu(grid, origin, res, U, n_slots=SLOTS, max_nodes=cap_n, max_edges=cap_e, max_log=cap_e, max_expand=150000, **KW)
    pr.setHelpers(helpers, -1)
    pr.setPoolRecycling(True)
    for rep in range(3):  # (repeats: chunks that come back in another order, helpers that lag behind)
        got = pr.planBatch(S, G)
        bad = [k for k in range(len(S)) if word(got[k]) != word(ref[k])]
        assert not bad, (rep, bad[:5], word(got[bad[0]]), word(ref[bad[0]]))
        assert all(np.array_equal(pr.getTraj(k).actions, ref_traj[k]) for k in range(len(S)))
    # the state spaces of a recycled batch are gone, and the library says so
    ctx = pr._ctx()
    n_rec, rs = C.c_uint64(), C.c_int32()
    assert ctx.lib.mplx_debug_query_records(ctx.h, 0, 0, None, C.byref(n_rec), C.byref(rs)) == _capi.ERR_ARG
    assert b"recycling" in ctx.lib.mplx_last_error(ctx.h)
    # without recycling the same pools are too small, and that is reported per query, not silently
    pr.setPoolRecycling(False)
    small = pr.planBatch(S, G)
    assert any(r.status == _capi.PLAN_POOL_FULL for r in small)


def test_a_single_plan_never_recycles_and_keeps_its_state_space():
    grid, origin, res, queries = batch(96, 4)
    U = mapgen.control_lattice(1.0, 1, True)
    mu, pl = util.make_gpu(grid, origin, res, U, **KW)
    pl.setPoolRecycling(True)
    s, g = queries[0]
    assert pl.plan(util.gpu_wp(s), util.gpu_wp(g))
    coords, pos, g, h, closed, opened = pl._nodes()  # (mplx_result_nodes: the single plan's state space is still there)
    assert len(g) == pl.getResult().n_nodes > 0 and int(closed.sum()) == pl.getResult().n_closed


def test_three_hundred_batches_on_one_context_repeat_across_the_epoch_wrap():
    """255 launch epochs, then the table is cleared and the count restarts: nothing an earlier batch left may ever look like a
    state of a later one (same queries every time -- the same keys land on the same slots)."""
    grid, origin, res, queries = batch(64, 12, seed=9)
    U = mapgen.control_lattice(1.0, 1, True)
    S = [util.gpu_wp(s) for s, g in queries]
    G = [util.gpu_wp(g) for s, g in queries]
    mu, pl = util.make_gpu(grid, origin, res, U, n_slots=12, max_nodes=1 << 20, max_edges=1 << 22, max_log=1 << 21, **KW)
    ref = [word(r) for r in pl.planBatch(S, G)]
    assert sum(w[0] == 0 for w in ref) >= 6
    for it in range(300):
        # alternate with a different batch (other keys, other queries' tags in the same slots) and with single plans
        if it % 3 == 1:
            pl.planBatch(G[:6], S[:6])
        elif it % 3 == 2:
            pl.plan(S[it % 12], G[it % 12])
            assert word(pl.getResult()) == ref[it % 12]
        got = [word(r) for r in pl.planBatch(S, G)]
        assert got == ref, it


def test_streamed_batches_recycle_like_blocking_ones():
    grid, origin, res, queries = batch(128, 64, seed=3)
    U = mapgen.control_lattice(1.0, 1, True)
    S = [util.gpu_wp(s) for s, g in queries]
    G = [util.gpu_wp(g) for s, g in queries]
    mu, pl = util.make_gpu(grid, origin, res, U, n_slots=4, max_nodes=1 << 23, max_edges=1 << 25, max_log=1 << 24, max_expand=100000, **KW)
    ref_r = pl.planBatch(S, G)
    ref = [word(r) for r in ref_r]
    total = sum(w[4] for w in ref)
    biggest = max(w[4] for w in ref)
    pl.setPoolRecycling(True)
    st = pl.stream(2)
    cap_n = 4 * (biggest + 2 * 32768)
    cap_e = 4 * (max(r.n_edges for r in ref_r) + 2 * 65536)
    st.configure(4, cap_n, cap_e, cap_e, -1, 0, 0, 4)
    tickets = [st.submit(S, G) for _ in range(2)]
    for rep in range(4):
        t = tickets.pop(0)
        assert [word(r) for r in st.wait(t)] == ref
        tickets.append(st.submit(S, G))
    for t in tickets:
        assert [word(r) for r in st.wait(t)] == ref
