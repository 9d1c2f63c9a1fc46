"""Per-phase table of the speculative kernel's batch loop, from the -DMPLX_LOOKUP_TIMERS build
(tools/build_kernel_variant.sh timers "-DMPLX_ONLY_ACC -DMPLX_LOOKUP_TIMERS -DMPLX_PHASE_TIMERS=1" "help spec").
  run tail   : query 1005 of the C4 stream (the one that runs into the 2 M cap) alone, helpers from the start
  run bulk   : the C4-ACC batch capped at 20 000 expansions per query, no helpers (256 compute units busy, no tail)
  parse FILE : the kernel's printf lines of one of the above -> mean cycles per batch and section, weighted by batches
usage: MPLX_LIB=build_tmp/libmplx_timers.so python tools/phase_table.py run tail > tail.txt; python tools/phase_table.py parse tail.txt"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# sections of the batch loop in program order: cyc2 slot -> name (mplx_spec.h MPLX_T2 marks; thread 0's clock)
SECTIONS = [
    (16, "top of loop: far link, evict, refill / top-up"),
    (17, "batch set-up left at the head"),
    (15, "selection: record prefetch issue + ranking"),
    (18, "selection: barrier behind the ranking"),
    (19, "selection: scatter to candidates / near set + barrier"),
    (20, "candidates' records arrive (wait for the prefetch)"),
    (21, "goal test, look-ahead record check, row request"),
    (-1, "expansion (get_succ: build, validate, sample) + unit scans"),
    (14, "lane keys to LDS"),
    (0, "(barrier of a batch-table clear: POT builds only)"),
    (1, "batch-table insert, row check, barrier"),
    (2, "table look-up: claim / record prefetch issued"),
    (3, "heuristic of the successor (or the helper's row)"),
    (4, "table look-up: probe loop (second dependent trip)"),
    (22, "candidate among the successors?"),
    (5, "barrier behind the look-up (arrival of the slowest lane)"),
    (6, "per-lane pre-commit values (tentative g, f, bucket, cut)"),
    (7, "barrier, cut set-up"),
    (8, "cut evaluation, closed flags"),
    (9, "barrier ahead of the commit"),
    (10, "parallel commit (scan, records, pushes)"),
    (11, "ordered commit (dependent batches only)"),
    (12, "barrier behind the commit"),
    (13, "counters, next batch's set-up, barrier"),
]


def run(mode):
    import numpy as np
    from mpl_ros_amd import mapgen
    from tests import util
    grid, origin, res, start, goal, _ = mapgen.benchmark_map(512)
    grid = np.ascontiguousarray(grid)
    U = mapgen.control_lattice(1.0, 1, True)
    queries = mapgen.c4_queries(grid, origin, res, 1024, rank=0)
    if mode == "tail":
        qi = int(os.environ.get("QI", "1005"))
        s, g = queries[qi]
        mu, pl = util.make_gpu(grid, origin, res, U, v_max=2.0, a_max=1.0, tol_pos=0.5, max_expand=2_000_000, max_nodes=1 << 24, max_edges=1 << 26, max_log=1 << 25)
        pl.setHelpers(-1, -1)
        for it in range(2):
            pl.plan(util.gpu_wp(s), util.gpu_wp(g))
            r = pl.getResult()
            cy = pl.queryCycles()
            print(f"RUN tail it {it} query {qi} expansions {r.n_expanded} kernel_ms {pl.lastKernelMs():.1f} batches {cy['batches']} dep {cy['dep_batches']} hits {cy['cache_hits']} expand_cyc {cy['expand']}", flush=True)
    else:
        cap = int(os.environ.get("CAP", "20000"))
        pools = mapgen.c4_pools(False, 1024, cap)
        mu, pl = util.make_gpu(grid, origin, res, U, v_max=2.0, a_max=1.0, tol_pos=0.5, max_expand=cap, n_slots=1024, max_nodes=pools["nodes"], max_edges=pools["edges"], max_log=pools["log"])
        pl.setHelpers(0, 0)
        starts = [util.gpu_wp(q[0]) for q in queries]
        goals = [util.gpu_wp(q[1]) for q in queries]
        for it in range(2):
            res_ = pl.planBatch(starts, goals)
            n = sum(r.n_expanded for r in res_)
            ex = sum(pl.queryCycles(k)["expand"] for k in range(len(res_)))
            nb = sum(pl.queryCycles(k)["batches"] for k in range(len(res_)))
            print(f"RUN bulk it {it} expansions {n} kernel_ms {pl.lastKernelMs():.1f} batches {nb} expand_cyc {ex}", flush=True)


def parse(path):
    txt = open(path, errors="replace").read()
    # (the kernel's printf lines are flushed when the process ends, i.e. behind both RUN lines: the second half is the last iteration)
    runs = re.findall(r"RUN \w+ it \d+ [^\n]*", txt)  # (the bulk run's kernel lines may interleave with it mid-line)
    run_line = runs[-1]
    lines = re.findall(r"cyc2 q(\d+) batches (\d+):((?: \d+){24})", txt)
    n_it = len(runs)
    lines = lines[len(lines) - len(lines) // max(n_it, 1):]
    body = txt
    tot_b, acc = 0, [0.0] * 24
    for _, b, vals in lines:
        b = int(b)
        v = [int(x) for x in vals.split()]
        tot_b += b
        for i in range(24):
            acc[i] += v[i] * b
    mean = [a / max(tot_b, 1) for a in acc]
    kv = dict(re.findall(r"(\w+) ([\d.]+)", run_line))
    n_exp, batches = float(kv["expansions"]), float(kv["batches"])
    expand = float(kv["expand_cyc"]) / batches
    print(run_line)
    print(f"queries parsed: {len(lines)}; batches {tot_b}; expansions per batch {n_exp / batches:.2f}")
    total = 0.0
    print(f"{'section':<66} {'cycles/batch':>12} {'cycles/expansion':>17}")
    for slot, name in SECTIONS:
        c = expand if slot < 0 else mean[slot]
        total += c
        print(f"{name:<66} {c:>12.0f} {c / (n_exp / batches):>17.0f}")
    print(f"{'sum (thread 0, barrier to barrier)':<66} {total:>12.0f} {total / (n_exp / batches):>17.0f}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        parse(sys.argv[2])
