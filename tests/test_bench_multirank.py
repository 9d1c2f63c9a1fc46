"""-m gpu: bench.py's multi-rank path on ONE GPU -- two ranks launched exactly as the driver launches them
(torch.distributed.run, one process per rank), both on device 0 (MPLX_BENCH_SHARE_GPU=1) with gloo standing in for
RCCL (which refuses two ranks on one device): map broadcast, the one-stream (strong) phase through run_sharded, the
throughput phase (a 2 x stream: the line's value), per-rank rows, one JSON line.  A small map and short queries: this checks the plumbing the
driver's 8-GPU run goes through, not performance."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_sharing_one_gpu_print_one_line_with_both_phases():
    env = dict(os.environ, MPLX_BENCH_SHARE_GPU="1", MPLX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--queries", "64", "--map", "128", "--max-expand", "20000", "--max-nodes", "60000", "--cpu-seconds", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    # the line = query throughput (weak scaling: 64 queries per GPU, a 128-query stream dealt over the two ranks) ...
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["metric"] == "node_expansions_per_s" and d["steps"] == 2
    assert d["config"]["queries_total"] == 128 and d["config"]["queries_per_gpu"] == 64
    assert len(d["per_rank"]) == 2 and all(p["queries"] == 64 and p["expansions_per_step"] > 0 for p in d["per_rank"])
    assert sum(p["expansions_per_step"] for p in d["per_rank"]) == d["expansions_per_step"]
    assert abs(d["value"] - d["expansions_per_step"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert d["map_setup_s"]["rccl_broadcast"] >= 0
    # ... and the strong-scaling measurement of the ONE 64-query stream rides along
    t = d["strong"]
    assert t["scaling"] == "strong" and t["queries_total"] == 64 and len(t["per_rank"]) == 2 and all(p["queries"] == 32 for p in t["per_rank"])
    assert sum(p["expansions_per_step"] for p in t["per_rank"]) == t["expansions_per_step"] < d["expansions_per_step"]
    assert t["tail_bound"]["longest_query_ms"] > 0 and t["ms_per_step"] >= t["tail_bound"]["longest_query_ms"] * 0.99  # a step cannot end before its longest query


def test_one_rank_over_rccl_takes_the_multi_rank_path():
    """The same path over the REAL back-end: backend "nccl" (= RCCL) refuses two ranks on one device, so the only way to run
    process-group set-up, the map broadcast from a device tensor, run_sharded's barriers / all_gather and the gather of the
    result rows through RCCL on a one-GPU box is a world of one rank (MPLX_BENCH_FORCE_DIST=1)."""
    env = dict(os.environ, MPLX_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--queries", "64", "--map", "128",
           "--max-expand", "20000", "--max-nodes", "60000", "--cpu-seconds", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["config"]["queries_total"] == 64 and len(d["per_rank"]) == 1
    assert d["strong"]["queries_total"] == 64 and d["strong"]["expansions_per_step"] == d["expansions_per_step"]  # one rank: the two phases hold the same stream
    assert "RCCL" in d["config"]["parallelism"] and d["map_setup_s"]["rccl_broadcast"] >= 0

