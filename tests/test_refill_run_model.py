"""CPU: a model of the OPEN structure's refill with run pulls (mpl_ros_amd/csrc/mplx_kernels.h refill / pull_fine_run).

Two things the kernel relies on, checked on a host model written from the kernel's rules (not from the reference: the bucketed OPEN list
is this back-end's own realisation of a priority queue under the order (f, g, id)):
  1. the run wave 0 selects -- non-empty buckets among the 64 from the lowest one on, while their counts add up to at most `target`
     and there are at most MAXB of them -- is a PREFIX of the non-empty buckets, starts with the lowest, is never empty, and holds more
     than `target` entries only when it is the lowest bucket alone (the single-bucket pull, which handles overflow by evictions);
  2. moving the near / far boundary to the end of ANY such run keeps the structure exact: with the near set = everything at or below the
     boundary bucket, the pop sequence equals a binary heap's on the same pushes (pushes never undercut the current minimum's bucket:
     a consistent heuristic; the kernel's demotion path for the other case is not modelled)."""
import heapq

import numpy as np

WINDOW = 64


def select_run(counts, b0, target, maxb):
    """the lanes' rule: lane l looks at bucket b0 + l; take = count > 0 and (l == 0 or (rank < maxb and inclusive prefix <= target))"""
    run, pre, rank = [], 0, 0
    for lane in range(WINDOW):
        b = b0 + lane
        c = counts[b] if b < len(counts) else 0
        pre += c
        if c > 0:
            if lane == 0 or (rank < maxb and pre <= target):
                run.append(b)
            rank += 1
    return run


def test_the_selected_run_is_a_prefix_of_the_non_empty_buckets():
    rng = np.random.default_rng(5)
    for trial in range(2000):
        nb = 1024
        density = rng.choice([0.02, 0.2, 0.7, 1.0])
        counts = np.where(rng.random(nb) < density, rng.integers(1, rng.choice([3, 40, 700]), nb), 0)
        nz = np.flatnonzero(counts)
        if len(nz) == 0:
            continue
        b0 = int(nz[0])
        target = int(rng.choice([0, 37, 128, 512]))
        maxb = int(rng.choice([1, 4, 8]))
        run = select_run(counts, b0, target, maxb)
        assert run and run[0] == b0 and len(run) <= max(1, maxb)
        assert run == [int(b) for b in nz[: len(run)]]                      # a prefix: no non-empty bucket is skipped
        assert all(b < b0 + WINDOW for b in run)
        total = int(counts[run].sum())
        assert total <= target or len(run) == 1                            # more than the target only as the single-bucket pull
        if len(run) < len(nz) and len(run) < maxb and nz[len(run)] < b0 + WINDOW and len(run) >= 1 and maxb > 1:
            assert total + int(counts[nz[len(run)]]) > target              # ... and it stops only where the next bucket would not fit


def test_any_run_boundary_keeps_the_pop_order():
    rng = np.random.default_rng(11)
    for trial in range(40):
        width = float(rng.choice([0.05, 0.3, 2.0]))
        target, maxb = int(rng.choice([4, 32, 512])), int(rng.choice([1, 4, 8]))
        nb = 4096
        far = [[] for _ in range(nb)]
        counts = np.zeros(nb, dtype=np.int64)
        near, cur0, heap, next_id = [], -1, [], 0
        bucket = lambda f: min(nb - 1, int(f / width))

        def push(f, g):
            nonlocal next_id
            e = (f, g, next_id)
            next_id += 1
            heapq.heappush(heap, e)
            b = bucket(f)
            if b <= cur0:
                near.append(e)
            else:
                far[b].append(e)
                counts[b] += 1

        push(0.0, 0.0)
        for step in range(3000):
            if not near:
                nz = np.flatnonzero(counts)
                if len(nz) == 0:
                    break
                run = select_run(counts, int(nz[0]), target, maxb)
                cur0 = run[-1]                                             # the boundary moves to the END of the run's last bucket
                for b in run:
                    near.extend(far[b])
                    far[b] = []
                    counts[b] = 0
                assert all(counts[: cur0 + 1] == 0)
            e = min(near)
            near.remove(e)
            assert e == heapq.heappop(heap), (trial, step)
            for _ in range(int(rng.integers(0, 4))):                       # successors: f never below the expanded entry's
                push(e[0] + float(rng.choice([0.0, 0.01, 0.4, 3.0])) * float(rng.random()), float(rng.random()))
