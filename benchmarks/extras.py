"""The other BASELINE configurations as extra keys of the driver's C4 line: c2 (config 2: one ACC query on the
256^3 map), c3 (config 3: one JRK query on the 512^3 map, cap 2 000 000), c5 (config 5: the 16-robot tick), lpa (the replanning
cycle at C2 size).  Each is the leg `bench.py --single ...` / `--config c5` / `--config lpa` runs, with few steps, reduced to its
headline numbers: value, kernel time, its own roofline, a single-thread CPU baseline and a parity check of the timed results.
A leg that fails is reported as {"error": ...}; it never takes the C4 line down."""
import copy
import time

from . import c4, c5, lpa
from .common import _log


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


def _single(args, n, lattice, steps, warmup, cpu_seconds, warmup_cap=0):
    a = copy.copy(args)
    a.single, a.map, a.lattice, a.steps, a.warmup, a.stream, a.cpu_seconds, a.max_expand, a.queries = True, n, lattice, steps, warmup, 0, cpu_seconds, 0, 1
    a.warmup_cap = warmup_cap
    a.dump_queries = ""
    return _compact(c4.run(a), ("speculation",))


def run(args):
    out = {}
    legs = (
        ("c2", lambda: _single(args, 256, "acc", 5, 1, 3.0)),
        ("c3", lambda: _single(args, 512, "jrk", 1, 1, 5.0, warmup_cap=20000)),
        ("c5", lambda: _c5(args)),
        ("lpa", lambda: _lpa(args)),
    )
    for name, fn in legs:
        t0 = time.perf_counter()
        _log(f"extra configuration {name}")
        try:
            out[name] = fn()
        except Exception as e:  # noqa: BLE001  (the C4 line stands on its own)
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        out[name]["leg_seconds"] = round(time.perf_counter() - t0, 1)
    return out


def _c5(args):
    a = copy.copy(args)
    a.steps, a.warmup, a.cpu_seconds, a.max_expand, a.c5_capped, a.helpers = 3, 1, 1.0, 0, False, -1
    d = c5.run(a)
    return _compact(d, ("tick_ms", "lookahead"))


def _lpa(args):
    a = copy.copy(args)
    a.steps, a.warmup, a.map, a.cpu_seconds = 2, 1, 256, 1.0
    d = lpa.run(a)
    return _compact(d, ("cycle", "lpa_vs_fresh_after_obstacle", "reroot"))
