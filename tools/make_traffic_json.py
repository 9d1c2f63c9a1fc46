#!/usr/bin/env python3
"""profiles/traffic.json from the rocprofv3 summaries of tools/profile_round.sh (profiles/summarize_rocprof.py output).
usage: make_traffic_json.py <summary_block.txt> <profile path to cite> [summary_bulk.txt] > profiles/traffic.json
Per counter: sum over the rows of one dispatch (a dispatch is identified by its duration), mean over the dispatches of the dominant
kernel; kernel_ms_trace = mean duration of the same kernel in the --kernel-trace pass.  bench.py attaches these numbers to its
roofline object only when kernel_ms_trace agrees with the kernel time it measures itself (3 %)."""
import json
import sys
from collections import defaultdict

KERNEL = "astar_spec_kernel"


def parse(path):
    sections, cur, mode = {}, None, None
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith("== "):
            cur = line[3:].split("/")[0]
            sections[cur] = {"stats": [], "counters": []}
            mode = None
        elif line.startswith("-- kernel stats"):
            mode = "stats"
        elif line.startswith("-- dispatches"):
            mode = "disp"
        elif line.startswith("-- counters"):
            mode = "counters"
        elif cur and line and mode in ("stats", "counters"):
            sections[cur][mode].append([x.strip() for x in line.split(" | ")])
    return sections


def per_launch(rows):
    """{counter: mean over dispatches of (sum over that dispatch's rows)}, and the dispatch durations (ms)"""
    acc = defaultdict(lambda: defaultdict(float))
    for r in rows:
        if KERNEL in r[0]:
            acc[r[1]][r[3]] += float(r[2])
    out, durs = {}, []
    for c, by in acc.items():
        out[c] = sum(by.values()) / len(by)
        durs += [float(d) * 1e-6 for d in by]
    return out, sorted(set(round(d, 3) for d in durs))


def main():
    S = parse(sys.argv[1])
    out = {"profile": sys.argv[2], "kernel": None}
    passes = []
    for name, sec in S.items():
        for r in sec["stats"]:
            if KERNEL in r[0] and name.startswith("trace"):
                out["kernel"] = r[0]
                out["kernel_ms_trace"] = float(r[3]) * 1e-3
                out["launches_in_trace"] = int(r[1])
        c, d = per_launch(sec["counters"])
        passes += d
        for k, v in c.items():
            out[k + ("_KB" if k in ("FETCH_SIZE", "WRITE_SIZE") else "")] = v
    out["kernel_ms_counter_passes"] = sorted(passes)
    if out.get("SQ_WAVE_CYCLES") and out.get("SQ_WAIT_ANY"):
        out["SQ_WAIT_ANY_over_WAVE_CYCLES"] = out["SQ_WAIT_ANY"] / out["SQ_WAVE_CYCLES"]
    out["note"] = ("rocprofv3 --kernel-trace --stats and --pmc FETCH_SIZE / WRITE_SIZE / SQ_* in separate passes (tools/profile_round.sh) of "
                   "`bench.py --steps 2 --warmup 1 --cpu-seconds 0 --stream 0`; values = mean per launch of the dominant kernel.  Counter units: KB of 1024 B; "
                   "calibration on this kernel's access patterns: profiles/r03m_counter_calibration.txt (a scattered 8-byte load counts its 64-byte line, a "
                   "scattered 8- or 16-byte store 32 bytes; no x2 correction: that rule holds for wide coalesced streams only).")
    if len(sys.argv) > 3:
        B = parse(sys.argv[3])
        bulk = {}
        for name, sec in B.items():
            c, d = per_launch(sec["counters"])
            bulk.update(c)
            if d:
                bulk.setdefault("kernel_ms", []).extend(d)
        out["bulk_phase"] = bulk
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
