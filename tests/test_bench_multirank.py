"""-m gpu: bench.py's multi-rank path on ONE GPU -- two ranks launched exactly as the driver launches them
(torch.distributed.run, one process per rank), both on device 0 (MPLX_BENCH_SHARE_GPU=1) with gloo standing in for
RCCL (which refuses two ranks on one device): map broadcast, run_sharded strong phase, the additional throughput
phase (a 2 x stream), per-rank rows, one JSON line.  A small map and short queries: this checks the plumbing the
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
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["metric"] == "node_expansions_per_s"
    assert d["config"]["queries_total"] == 64 and d["config"]["queries_rank0"] == 32
    assert len(d["per_rank"]) == 2 and all(p["queries"] == 32 and p["expansions_per_step"] > 0 for p in d["per_rank"])
    assert sum(p["expansions_per_step"] for p in d["per_rank"]) == d["expansions_per_step"]
    assert d["map_setup_s"]["rccl_broadcast"] >= 0 and d["tail_bound"]["longest_query_ms"] > 0
    assert d["ms_per_step"] >= d["tail_bound"]["longest_query_ms"] * 0.99  # a step cannot end before its longest query
    t = d["throughput"]
    assert t["scaling"] == "weak" and t["queries_total"] == 128 and t["queries_per_gpu"] == 64 and len(t["per_rank"]) == 2
    assert sum(p["expansions_per_step"] for p in t["per_rank"]) == t["expansions_per_step"] > d["expansions_per_step"]
