"""The blocking C4 batches must reproduce bit for bit while something else writes to the device's memory.

Round 4 found the C4-ACC batch differing in 4 queries of 10 batches under a background fill load (states created twice, OPEN lists
running dry) and closed most of it with resolved table claims and self-validating look-ahead rows; round 5 found what was underneath --
the commit read the pool counters behind the scan's barrier, where a wave that had fallen behind (a store queue under back-pressure is
what makes a wave fall behind) picked up the updated totals -- and the jerk batch showing it on a quiet device.  No oracle needed: a
batch is compared with itself.  Runs as a subprocess: the probe needs torch on the device for its background load, and torch's
bundled HIP runtime only initialises in a process where it comes FIRST (tools/fill_load_probe.py).
(File name: runs last -- a failure here must not hide the parity tests.)"""
import gc
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("lattice,batches", [("acc", 4), ("jrk", 8)])
def test_c4_batch_repeats_under_a_background_fill_load(lattice, batches):
    gc.collect()  # (the batch's pools take ~ 130 GB, the background buffer 16 GB: nothing of earlier tests should linger)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fill_load_probe.py"), str(batches), "fill", lattice], capture_output=True, text=True, timeout=420,
                       cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["fill_rounds"] >= 2, out  # (the load really ran next to the batches)
    assert out["mismatching_queries"] == 0, out
