#!/usr/bin/env python3
"""Table of the kernels' resources from the remarks of `make -C mpl_ros_amd/csrc resource-usage` (hipcc -Rpass-analysis=kernel-resource-usage).
usage: make -C mpl_ros_amd/csrc -j8 resource-usage > raw.txt 2>&1; python tools/resource_usage_table.py raw.txt > profiles/<tag>_resource_usage.txt"""
import re
import subprocess
import sys

FIELDS = [("VGPRs", "VGPR"), ("VGPRs Spill", "VGPR spill"), ("SGPRs Spill", "SGPR spill"), ("ScratchSize [bytes/lane]", "scratch B"), ("LDS Size [bytes/block]", "LDS B"),
          ("Occupancy [waves/SIMD]", "waves/SIMD")]


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return out[:len(names)]
    except Exception:  # noqa: BLE001
        return names


def main():
    rows, cur = [], None
    for line in open(sys.argv[1]):
        m = re.search(r"remark: [^:]*:\d+:\d+: (.*?) \[-Rpass-analysis", line) or re.search(r"remark: (.*?) \[-Rpass-analysis", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.rsplit(":", 1)
            cur[k.strip()] = v.strip()
    names = demangle([r["name"] for r in rows])
    seen = set()
    print("kernel resource usage of the shipped libmplx.so (hipcc -Rpass-analysis=kernel-resource-usage, gfx950; make -C mpl_ros_amd/csrc resource-usage; tools/resource_usage_table.py).")
    print("template arguments of astar_spec_kernel: <unit lanes, units, control (3 ACC / 7 JRK), batch-table size, near-set size, HELP, POT, YAW, FILTER>; its units are compiled at -O2 (FLAGS_SPEC)\n")
    print(f"{'kernel':112s} " + " ".join(f"{h:>10s}" for _, h in FIELDS))
    for r, n in zip(rows, names):
        n = re.sub(r"^void ", "", n)
        n = re.sub(r"\(mplx::SearchParams.*$|\(.*\)$", lambda m: "" if "SearchParams" in m.group(0) else m.group(0), n).replace("mplx::", "")
        n = re.sub(r"\((int)\)", "", n)
        if n in seen:
            continue
        seen.add(n)
        print(f"{n[:112]:112s} " + " ".join(f"{r.get(k, '?'):>10s}" for k, _ in FIELDS))


if __name__ == "__main__":
    main()
