"""The blocking C4-ACC batch must reproduce bit for bit while something else writes to the device's memory (round 4).

A posted agent-scope store that a loaded memory system holds back used to let a look-up meet its own table claim before the
entry landed (a state created twice) and a leader read a look-ahead row before its contents (wrong heuristics): invisible on a
quiet device, 4 differing queries in 10 batches under this load.  No oracle needed: the batch is compared with itself.
(File name: runs last -- a failure here must not hide the parity tests.)"""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_c4_acc_batch_repeats_under_a_background_fill_load():
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()  # (the batch's pools take ~ 130 GB, the background buffer 16 GB: nothing of earlier tests should linger)
    spec = importlib.util.spec_from_file_location("r04_jitter_probe", os.path.join(ROOT, "tools", "r04_jitter_probe.py"))
    probe = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(probe)
    out = probe.run(4, "fill")
    assert out["fill_rounds"] >= 4, out  # (the load really ran next to the batches)
    assert out["mismatching_queries"] == 0, out
