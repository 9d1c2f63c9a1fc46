"""The other BASELINE configurations as extra keys of the driver's C4 line: c2 (config 2: one ACC query on the
256^3 map), c3 (config 3: one JRK query on the 512^3 map, cap 2 000 000), c5 (config 5: the 16-robot tick), lpa (the replanning
cycle at C2 size), plpa (LPA* on the moving-obstacle planner: the replanner node's flow).  Each is the leg `bench.py --single ...` / `--config c5` / `--config lpa` runs -- in a process of its own -- with few steps, reduced to its
headline numbers: value, kernel time, its own roofline, a single-thread CPU baseline and a parity check of the timed results.
A leg that fails is reported as {"error": ...}; it never takes the C4 line down."""
import json
import os
import subprocess
import sys
import time

from .common import ROOT, _log


def _compact(d, extra=()):
    keep = ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "higher_is_better", "expansions_per_step", "plan_status_counts", "parity_sample", "vs_cpu_single_thread") + tuple(extra)
    out = {k: d[k] for k in keep if k in d}
    out["workload"] = d["config"]["workload"]
    r = d.get("roofline", {})
    out["roofline"] = {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "algorithmic_bytes_per_launch") if k in r}
    cb = d.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample", "build", "error", "tick_ms") if k in cb}
        if cb.get("value") and d.get("metric") == "node_expansions_per_s":
            out["vs_cpu_single_thread"] = d["value"] / cb["value"]
    return out


def _child(argv, timeout_s):
    """One leg as a process of its own (`python bench.py ...`): a leg that dies -- a device fault ends its process -- costs the line
    one key, not the line; its ONE JSON line is parsed from stdout."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + [str(a) for a in argv]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError(f"exit code {r.returncode}: {(r.stderr or r.stdout)[-400:]}")
    return json.loads(lines[-1])


def run(args):
    out = {}
    legs = (
        ("c2", ["--single", "--map", 256, "--lattice", "acc", "--steps", 5, "--warmup", 1, "--cpu-seconds", 3, "--stream", 0, "--extras", 0], ("speculation",), 90),
        ("c3", ["--single", "--map", 512, "--lattice", "jrk", "--steps", 1, "--warmup", 1, "--warmup-cap", 20000, "--cpu-seconds", 5, "--stream", 0, "--extras", 0], ("speculation",), 120),
        ("c5", ["--config", "c5", "--steps", 3, "--warmup", 1, "--cpu-seconds", 1], ("tick_ms", "lookahead"), 90),
        ("lpa", ["--config", "lpa", "--map", 256, "--steps", 2, "--warmup", 1, "--cpu-seconds", 1], ("cycle", "lpa_vs_fresh_after_obstacle"), 90),
        ("plpa", ["--config", "plpa", "--steps", 3, "--warmup", 1, "--cpu-seconds", 1], ("replans", "fresh_plan_wall_ms_mean", "repair_vs_fresh", "all_costs_equal_fresh"), 60),
    )
    for name, argv, extra, timeout_s in legs:
        t0 = time.perf_counter()
        _log(f"extra configuration {name}")
        try:
            out[name] = _compact(_child(argv, timeout_s), extra)
        except Exception as e:  # noqa: BLE001  (the C4 line stands on its own)
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        out[name]["leg_seconds"] = round(time.perf_counter() - t0, 1)
    return out
