"""CPU: the evidence bench.py attaches must be what the committed rocprofv3 summary says (round 4 shipped counters of a
kernel that was not the one timed).  profiles/traffic.json names the summary it was generated from; regenerating it from that file
must give the same numbers, its kernel time must be the --kernel-trace average of the dominant kernel, and the driver-form bench
line committed next to it must agree with that kernel time within the 3 % bench.py itself demands before it attaches counters."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("make_traffic_json", os.path.join(ROOT, "tools", "make_traffic_json.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_traffic_json_is_the_committed_summary_and_agrees_with_the_bench_line():
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    summary = os.path.join(ROOT, t["profile"])
    assert os.path.exists(summary), t["profile"]
    m = _tool()
    S = m.parse(summary)
    trace = [r for name, sec in S.items() if name.startswith("trace") for r in sec["stats"] if m.KERNEL in r[0]]
    assert trace, "no --kernel-trace section of the dominant kernel in the committed summary"
    assert abs(float(trace[0][3]) * 1e-3 - t["kernel_ms_trace"]) < 1e-6 and int(trace[0][1]) == t["launches_in_trace"] >= 2
    for name, sec in S.items():
        vals, _ = m.per_launch(sec["counters"])
        for k, v in vals.items():
            key = k + ("_KB" if k in ("FETCH_SIZE", "WRITE_SIZE") else "")
            assert abs(t[key] - v) <= 1e-9 * max(1.0, abs(v)), (name, k)
    assert t["FETCH_SIZE_KB"] > 0 and t["WRITE_SIZE_KB"] > 0 and t["SQ_INSTS_VALU"] > 0
    # every launch of the counter passes ran the same kernel at the same speed (a pass that perturbed it would not describe it)
    assert max(t["kernel_ms_counter_passes"]) < 1.03 * t["kernel_ms_trace"] and min(t["kernel_ms_counter_passes"]) > 0.97 * t["kernel_ms_trace"]
    # the committed driver-form bench line of the same run
    tag = os.path.basename(t["profile"]).split("_")[0]
    bench = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench_default.json")))
    k_ms = bench["roofline"]["kernel_ms"]
    assert abs(k_ms - t["kernel_ms_trace"]) <= 0.03 * k_ms, (k_ms, t["kernel_ms_trace"])
    assert bench["roofline"]["algorithmic_bytes_per_launch"] == 273544792870 and bench["steps"] == 20 and bench["warmup"] == 5
    assert abs(bench["roofline"]["frac"] - bench["roofline"]["achieved"] / 8000.0) < 1e-12
