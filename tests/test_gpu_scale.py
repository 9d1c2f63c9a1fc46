"""-m gpu: parity at the BASELINE.json configuration sizes, through the C-ABI.

  C2  256^3 random-box map, the 8(d) query, 27-primitive ACC lattice, run to the goal: the whole plan is
      compared (expansion order hash, states, predecessor lists, counters, path actions / node ids / waypoint
      states, cost) -- util.compare_plan.
  C3  512^3 map, 125-primitive JRK lattice, both sides capped at the SAME number of expansions: same fields.
  C4  the bench's own 1024-query batch on the 512^3 map (ACC cap 2 000 000, JRK cap 20 000): a sample of
      >= 32 queries that always contains the longest one is replayed on the CPU oracle (threads, one
      read-only map) and compared on status, expansions, order hash, states, edges, voxel reads, cost and
      the path's actions / node ids.
Everything is bit-exact (integers and f64 alike)."""
import gc
import threading

import numpy as np
import pytest

from mpl_ros_amd import mapgen
from oracle import orc
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def map512():
    grid, origin, res, start, goal, _ = mapgen.benchmark_map(512)
    return grid, origin, res, start, goal


def test_c2_full_query_acc_256():
    grid, origin, res, start, goal, _ = mapgen.benchmark_map(256)
    U = mapgen.control_lattice(1.0, 1, True)
    kw = dict(v_max=2.0, a_max=1.0, tol_pos=0.5)
    P = util.make_oracle(grid, origin, res, orc.ACC, U, **kw)
    mu, pl = util.make_gpu(grid, origin, res, U, max_nodes=1 << 22, max_edges=1 << 24, max_log=1 << 23, **kw)
    r, c = util.compare_plan(P, pl, (start, (0, 0, 0)), (goal,), orc.ACC)
    assert r.status == 0 and r.n_expanded > 10000
    print(f"C2: expanded {r.n_expanded} states {r.n_nodes} edges {r.n_edges} cost {r.cost} kernel {pl.lastKernelMs():.1f} ms")


def test_c2_query_does_not_depend_on_the_open_bucket_width():
    """The OPEN structure's bucket width is a speed knob only: the C2 query at 1 / 30 of the default width (sparse fine buckets: every
    refill pulls a run of them in one walk -- mplx_kernels.h pull_fine_run --, 64-bucket windows with gaps, a coarse bucket activated every
    few batches), at the default, and at 30 times the default (dense buckets pulled alone, near-set evictions) returns the same words,
    expansion order included; one of them is compared with the oracle in full by the test above."""
    grid, origin, res, start, goal, _ = mapgen.benchmark_map(256)
    U = mapgen.control_lattice(1.0, 1, True)
    kw = dict(v_max=2.0, a_max=1.0, tol_pos=0.5)
    mu, pl = util.make_gpu(grid, origin, res, U, max_nodes=1 << 22, max_edges=1 << 24, max_log=1 << 23, **kw)
    words, refills = [], []
    for width in (1.0, 0.0, 30.0, 900.0):  # (w dt = 10: a tenth of an edge cost, the default of 3 edge costs, 3, 90)
        pl.setBucketWidth(width)
        assert pl.plan(util.gpu_wp(start), util.gpu_wp(goal))
        r = pl.getResult()
        words.append((r.status, r.traj_len, r.cost, r.n_expanded, r.n_nodes, r.n_edges, r.n_succ_finite, r.voxel_reads, r.n_push, r.expand_hash))
        refills.append((width, r.n_refill, r.n_evict, round(pl.lastKernelMs(), 1)))
    print("C2 (width, refills, evictions, kernel ms):", refills)
    assert all(w == words[0] for w in words), words
    assert words[0][3] > 10000


@pytest.mark.parametrize("cap", [250_000])
def test_c3_single_query_jrk_512_equal_cap(map512, cap):
    grid, origin, res, start, goal = map512
    U = mapgen.control_lattice(1.0, 2, True)
    assert U.shape[0] == 125
    kw = dict(v_max=2.0, a_max=1.0, j_max=1.0, tol_pos=0.5, max_expand=cap)
    P = util.make_oracle(grid, origin, res, orc.JRK, U, **kw)
    mu, pl = util.make_gpu(grid, origin, res, U, max_nodes=1 << 23, max_edges=1 << 25, max_log=1 << 24, **kw)
    r, c = util.compare_plan(P, pl, (start, (0, 0, 0), (0, 0, 0)), (goal,), orc.JRK)
    assert r.n_expanded == cap and r.status == 3
    print(f"C3: expanded {r.n_expanded} states {r.n_nodes} edges {r.n_edges} voxel reads {r.voxel_reads} kernel {pl.lastKernelMs():.1f} ms")


def test_c3_prefix_does_not_depend_on_the_open_bucket_width(map512):
    """The same for the 125-input jerk lattice (four 128-lane units; its near set has room for runs of about 500 entries next to the
    504 an iteration reserves): the first 60 000 expansions of the C3 query at a tenth of the default width, the default (half an edge
    cost), 10 and 200 edge costs -- the same words, expansion order included."""
    grid, origin, res, start, goal = map512
    U = mapgen.control_lattice(1.0, 2, True)
    kw = dict(v_max=2.0, a_max=1.0, j_max=1.0, tol_pos=0.5, max_expand=60_000)
    mu, pl = util.make_gpu(grid, origin, res, U, max_nodes=1 << 22, max_edges=1 << 25, max_log=1 << 23, **kw)
    words, refills = [], []
    for width in (0.5, 0.0, 100.0, 2000.0):  # (w dt = 10)
        pl.setBucketWidth(width)
        pl.plan(util.gpu_wp(start, control=orc.JRK), util.gpu_wp(goal, control=orc.JRK))
        r = pl.getResult()
        words.append((r.status, r.traj_len, r.cost, r.n_expanded, r.n_nodes, r.n_edges, r.n_succ_finite, r.voxel_reads, r.n_push, r.expand_hash))
        refills.append((width, r.n_refill, r.n_evict, round(pl.lastKernelMs(), 1)))
    print("C3 prefix (width, refills, evictions, kernel ms):", refills)
    assert all(w == words[0] for w in words), words
    assert words[0][0] == 3 and words[0][3] == 60_000


def test_c3_single_query_jrk_512_full_cap(map512):
    """BASELINE config 3 at its stated size: the 125-input jerk lattice on the 512^3 map, capped at 2 000 000
    expansions on both sides (deep OPEN lists, far-bucket pulls, the helper cache of the JRK kernel -- the part of
    the search the 250 000-expansion test never reaches).  The oracle needs about a minute on one core.  Compared:
    status, expansion count and order (expand_hash folds every expanded node id in order), states created,
    predecessor records, closed set, successor / primitive / voxel-read counters.  The full predecessor-list dump
    (a few hundred million records) is left to the equal-cap test above."""
    grid, origin, res, start, goal = map512
    cap = 2_000_000
    U = mapgen.control_lattice(1.0, 2, True)
    kw = dict(v_max=2.0, a_max=1.0, j_max=1.0, tol_pos=0.5, max_expand=cap)
    P = util.make_oracle(grid, origin, res, orc.JRK, U, **kw)
    pools = mapgen.c4_pools(True, 1, cap)
    mu, pl = util.make_gpu(grid, origin, res, U, max_nodes=pools["nodes"], max_edges=pools["edges"], max_log=pools["log"], **kw)
    ok = pl.plan(util.gpu_wp(start, control=orc.JRK), util.gpu_wp(goal, control=orc.JRK))
    r = pl.getResult()
    P.reset_counters()
    st = P.plan(orc.waypoint(start, control=orc.JRK), orc.waypoint(goal, control=orc.JRK))
    c = P.counters()
    ids, _ = P.expanded()
    assert r.status == st and ok == (st == orc.OK)
    assert r.n_expanded == c["n_expansions"] == len(ids)
    assert r.n_expanded == cap or st == orc.OK
    assert r.expand_hash == util.expand_hash(ids)
    assert r.n_nodes == P.num_nodes() and r.n_closed == P.num_closed()
    assert r.n_edges == c["n_succ_finite"] == r.n_succ_finite
    assert r.n_succ == c["n_succ"] and r.n_primitives == c["n_primitives"]
    assert r.voxel_reads == c["n_voxel_reads"]
    assert r.n_reopen == c["n_reopen"]
    if st == orc.OK:
        assert r.cost == P.traj_cost
    print(f"C3 full cap: status {r.status} expanded {r.n_expanded} states {r.n_nodes} edges {r.n_edges} voxel reads {r.voxel_reads} "
          f"kernel {pl.lastKernelMs():.0f} ms")


def test_c3_repeated_plans_are_identical(map512):
    """The BASELINE C3 query (helper-assisted 125-input kernel, 2 000 000 expansions) planned three times: every result
    word must repeat.  A race between a leader's waves, or between a leader and its helpers, shows here as a different
    state count or expansion-order hash -- this is how the stale-claim hazard of the table probe was found (a slot's
    claim seen through the compute unit's L1 after the slot had received its entry made the leader create a state twice,
    in about half of the runs once the control inputs no longer churned the L1; mplx_spec.h: the look-up re-reads a claim
    with the query's tag past the L1)."""
    grid, origin, res, start, goal = map512
    U = mapgen.control_lattice(1.0, 2, True)
    kw = dict(v_max=2.0, a_max=1.0, j_max=1.0, tol_pos=0.5, max_expand=2_000_000)
    pools = mapgen.c4_pools(True, 1, 2_000_000)
    mu, pl = util.make_gpu(grid, origin, res, U, max_nodes=pools["nodes"], max_edges=pools["edges"], max_log=pools["log"], **kw)
    seen = set()
    for it in range(3):
        pl.plan(util.gpu_wp(start, control=orc.JRK), util.gpu_wp(goal, control=orc.JRK))
        r = pl.getResult()
        seen.add((r.status, r.n_expanded, r.expand_hash, r.n_nodes, r.n_edges, r.voxel_reads, r.n_succ, r.n_succ_finite))
    assert len(seen) == 1, seen


def _cpu_replay(grid, origin, res, control, U, kw, queries, idx, threads=16):
    """Plan queries[i] for i in idx on the oracle; returns {i: dict of results}.  One shared read-only map."""
    out, lock, todo = {}, threading.Lock(), list(idx)

    def work():
        P = orc.Planner()
        P.set_map_shared(grid, origin, res)
        P.set_config(control, U, **kw)
        while True:
            with lock:
                if not todo:
                    return
                i = todo.pop(0)
            s, g = queries[i]
            P.reset_counters()
            st = P.plan(orc.waypoint(s, control=control), orc.waypoint(g, control=control))
            ids, _ = P.expanded()
            c = P.counters()
            tr = P.traj() if st == orc.OK else None
            co, po, ao = P.edges()
            with lock:
                out[i] = dict(status=st, n_expanded=len(ids), hash=util.expand_hash(ids), n_nodes=P.num_nodes(), n_edges=len(co),
                              reads=c["n_voxel_reads"], n_succ=c["n_succ"], n_fin=c["n_succ_finite"], cost=P.traj_cost,
                              actions=None if tr is None else tr["actions"], node_ids=None if tr is None else tr["node_ids"])

    ths = [threading.Thread(target=work) for _ in range(min(threads, len(todo)))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return out


@pytest.mark.parametrize("lattice", ["acc", "jrk"])
def test_c4_batch_sample_matches_the_oracle(map512, lattice):
    grid, origin, res, _, _ = map512
    grid = np.ascontiguousarray(grid)
    nq = 1024
    jrk = lattice == "jrk"
    control = orc.JRK if jrk else orc.ACC
    U = mapgen.control_lattice(1.0, 2 if jrk else 1, True)
    cap = 20000 if jrk else 2_000_000
    kw = dict(v_max=2.0, a_max=1.0, tol_pos=0.5, max_expand=cap)
    if jrk:
        kw["j_max"] = 1.0
    queries = mapgen.c4_queries(grid, origin, res, nq, rank=0)
    pools = mapgen.c4_pools(jrk, nq, cap)
    mu, pl = util.make_gpu(grid, origin, res, U, n_slots=768 if jrk else 1024, max_nodes=pools["nodes"], max_edges=pools["edges"],
                           max_log=pools["log"], **kw)
    R = pl.planBatch([util.gpu_wp(s, control=control) for s, g in queries], [util.gpu_wp(g, control=control) for s, g in queries])
    assert all(r.status in (0, 1, 3) for r in R), np.bincount([r.status for r in R])
    ne = np.array([r.n_expanded for r in R])
    longest = int(np.argmax(ne))
    sample = sorted(set([longest] + list(range(31)) + [int(np.argsort(ne)[len(ne) // 2])]))
    cpu = _cpu_replay(grid, origin, res, control, U, kw, queries, sample)
    for i in sample:
        r, c = R[i], cpu[i]
        assert r.status == c["status"], (i, r.status, c["status"])
        assert r.n_expanded == c["n_expanded"] and r.expand_hash == c["hash"], i
        assert r.n_nodes == c["n_nodes"] and r.n_edges == c["n_edges"], i
        assert r.voxel_reads == c["reads"] and r.n_succ == c["n_succ"] and r.n_succ_finite == c["n_fin"], i
        if c["status"] == orc.OK:
            assert r.cost == c["cost"], i  # bit-exact f64
            tg = pl.getTraj(i)
            assert np.array_equal(tg.actions, c["actions"]) and np.array_equal(tg.node_ids, c["node_ids"]), i
        else:
            assert np.isinf(r.cost)
    print(f"C4-{lattice}: {len(sample)} of {nq} queries replayed on the CPU (longest: query {longest}, {ne[longest]} expansions); "
          f"batch {ne.sum()} expansions in {pl.lastKernelMs():.0f} ms")


def test_c4_jrk_batch_is_the_same_with_and_without_helpers(map512):
    """The 1024-query jerk batch (<128,4,JRK,help>: units of 128 lanes = two waves) planned without helper workgroups and three
    times with them: every result word of every query must be identical.  Round 5: 20-60 queries differed from run to run -- the
    ordered commit read the pool counters behind the scan's barrier, where a wave that had fallen behind the publishing one picked
    up the UPDATED totals (its records landed past them: holes of unwritten node records, states overwritten by the next batch).
    The oracle sample of the test above compares 33 of the 1024 queries and could miss all of them; this one compares all."""
    grid, origin, res, _, _ = map512
    grid = np.ascontiguousarray(grid)
    nq, cap = 1024, 20000
    U = mapgen.control_lattice(1.0, 2, True)
    kw = dict(v_max=2.0, a_max=1.0, j_max=1.0, tol_pos=0.5, max_expand=cap)
    queries = mapgen.c4_queries(grid, origin, res, nq, rank=0)
    pools = mapgen.c4_pools(True, nq, cap)
    mu, pl = util.make_gpu(grid, origin, res, U, n_slots=768, max_nodes=pools["nodes"], max_edges=pools["edges"], max_log=pools["log"], **kw)
    S = [util.gpu_wp(s, control=orc.JRK) for s, g in queries]
    G = [util.gpu_wp(g, control=orc.JRK) for s, g in queries]
    word = lambda r: (r.status, r.traj_len, r.cost, r.n_expanded, r.n_nodes, r.n_edges, r.n_succ_finite, r.voxel_reads, r.n_push, r.expand_hash)
    pl.setHelpers(0, -1)
    ref = [word(r) for r in pl.planBatch(S, G)]
    assert pl.kernelName() == "astar_spec_kernel<128,4,JRK>"
    pl.setHelpers(-1, -1)
    for it in range(3):
        got = [word(r) for r in pl.planBatch(S, G)]
        assert pl.kernelName() == "astar_spec_kernel<128,4,JRK,help>"
        bad = [i for i, (a, b) in enumerate(zip(got, ref)) if a != b]
        assert not bad, (it, len(bad), bad[:8])


def test_c4_acc_batch_repeats_blocking_and_streamed(map512):
    """Repeatability at bench size (DESIGN: memory-ordering contract): the 1024-query C4-ACC batch planned
    twice blocking (reserved helpers) and four times
    through a two-lane stream with a helper limit (helpers attach and leave in the middle of queries, two launches share
    the machine): all 1024 result tuples of every run must be identical.  No oracle involved: any race between a leader's
    waves, a leader and its helpers, or two launches shows as a differing state count / order hash."""
    grid, origin, res, _, _ = map512
    grid = np.ascontiguousarray(grid)
    nq, cap = 1024, 2_000_000
    U = mapgen.control_lattice(1.0, 1, True)
    kw = dict(v_max=2.0, a_max=1.0, tol_pos=0.5, max_expand=cap)
    queries = mapgen.c4_queries(grid, origin, res, nq, rank=0)
    pools = mapgen.c4_pools(False, nq, cap, per_q=420_000)
    mu, pl = util.make_gpu(grid, origin, res, U, n_slots=1024, max_nodes=pools["nodes"], max_edges=pools["edges"], max_log=pools["log"], **kw)
    S = [util.gpu_wp(s) for s, g in queries]
    G = [util.gpu_wp(g) for s, g in queries]
    word = lambda r: (r.status, r.traj_len, r.cost, r.n_expanded, r.n_nodes, r.n_edges, r.n_succ_finite, r.voxel_reads, r.n_push, r.expand_hash)
    ref = [word(r) for r in pl.planBatch(S, G)]
    ms0 = pl.lastKernelMs()
    assert [word(r) for r in pl.planBatch(S, G)] == ref
    pl.releasePools()
    st = pl.stream(2)
    st.configure(256, pools["nodes"], pools["edges"], pools["log"], -1, 0, 1 << 24, 32)
    tickets = [st.submit(S, G), st.submit(S, G)]
    bad = 0
    for k in range(4):
        t = tickets.pop(0)
        bad += sum(1 for r, w in zip(st.wait(t), ref) if word(r) != w)
        if k < 2:
            tickets.append(st.submit(S, G))
    st.close()
    print(f"C4-ACC repeat: blocking {ms0:.0f} ms, 4 streamed batches, mismatching tuples {bad}")
    assert bad == 0


@pytest.mark.parametrize("nq", [1, 40, 600])
def test_helpers_leave_every_result_unchanged(nq):
    """Helper workgroups (mplx_set_helpers): a batch smaller than the machine gets extra workgroups that help from the
    start, a larger one has its leaders turn into helpers as the queries run out; off / on / a reserved share / four per leader must give
    identical plans -- counters, expansion-order hash, cost, path -- and the first few are replayed on the CPU oracle."""
    grid, origin, res, start, goal, _ = mapgen.benchmark_map(128)
    grid = np.ascontiguousarray(grid)
    U = mapgen.control_lattice(1.0, 1, True)
    cap = 20000
    kw = dict(v_max=2.0, a_max=1.0, tol_pos=0.5, max_expand=cap)
    queries = mapgen.c4_queries(grid, origin, res, nq, rank=3, min_dist=6.0)
    pools = mapgen.c4_pools(False, nq, cap, per_q=200_000 if nq == 1 else 100_000)  # (every query takes at least one 32 768-state chunk of the shared pool)
    S = [util.gpu_wp(s, control=orc.ACC) for s, g in queries]
    G = [util.gpu_wp(g, control=orc.ACC) for s, g in queries]
    runs = {}
    gc.collect()
    for name, (per, reserved) in {"off": (0, -1), "on": (2, -1), "reserved": (2, 64), "four": (4, -1)}.items():
        mu, pl = util.make_gpu(grid, origin, res, U, n_slots=min(nq, 1024), max_nodes=pools["nodes"], max_edges=pools["edges"], max_log=pools["log"], **kw)
        pl.setHelpers(per, reserved)
        R = pl.planBatch(S, G)
        runs[name] = [(r.status, r.n_expanded, r.expand_hash, r.n_nodes, r.n_edges, r.voxel_reads, r.n_succ, r.n_succ_finite, r.cost,
                       tuple(pl.getTraj(i).actions.tolist()) if r.status == 0 else None) for i, r in enumerate(R)]
        st = pl.helperStats()
        assert st["helpers_gave_up"] == 0
        if per:
            assert st["queries_done"] == nq
        hits = sum(pl.queryCycles(i)["cache_hits"] for i in range(nq))
        print(f"helpers {name}: batch of {nq}, {sum(r.n_expanded for r in R)} expansions, kernel {pl.lastKernelMs():.1f} ms, cache hits {hits}, {st}")
        if name == "off":
            assert hits == 0
        del mu, pl, R
        gc.collect()  # (the context and its pools go with the planner)
    assert runs["on"] == runs["off"] and runs["reserved"] == runs["off"] and runs["four"] == runs["off"]  # (four: the helpers of a leader split its list four ways)
    sample = list(range(min(nq, 4)))
    cpu = _cpu_replay(grid, origin, res, orc.ACC, U, kw, queries, sample)
    for i in sample:
        st, ne, hh, nn, ned, reads, nsucc, nfin, cost, acts = runs["on"][i]
        c = cpu[i]
        assert (st, ne, hh, nn, ned, reads, nsucc, nfin) == (c["status"], c["n_expanded"], c["hash"], c["n_nodes"], c["n_edges"], c["reads"], c["n_succ"], c["n_fin"]), i
        if st == 0:
            assert cost == c["cost"] and acts == tuple(np.asarray(c["actions"]).tolist())
