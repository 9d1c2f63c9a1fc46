#!/usr/bin/env python3
"""Turns rocprofv3 (rocpd sqlite) outputs into the small text summaries committed under profiles/.

usage: summarize_rocprof.py <dir with */<name>_results.db> > profiles/<round>_<what>.txt
Prints, per database: the --kernel-trace --stats table (calls, total, average duration per kernel)
and, for --pmc runs, the counter value per kernel dispatch.
"""
import glob
import os
import sqlite3
import sys


def main(root):
    for db_path in sorted(glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True)):
        db = sqlite3.connect(db_path)
        cur = db.cursor()
        print(f"== {os.path.relpath(db_path, root)}")
        print("-- kernel stats (us): name | calls | total | average | %")
        for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            print(f"{r[0]} | {r[1]} | {r[2]:.0f} | {r[3]:.0f} | {r[4]:.4f}")
        print("-- dispatches: kernel | grid | wg | lds | scratch | vgpr | sgpr | duration_ns")
        for r in cur.execute("select name,grid_x,workgroup_x,lds_size,scratch_size,vgpr_count,sgpr_count,duration from kernels"):
            print(" | ".join(str(x) for x in r))
        tabs = [t[0] for t in cur.execute("select name from sqlite_master where type in ('table','view')")]
        if "counters_collection" in tabs:
            print("-- counters: kernel | counter | value | duration_ns")
            for r in cur.execute("select kernel_name,counter_name,value,duration from counters_collection"):
                print(" | ".join(str(x) for x in r))
        print()


if __name__ == "__main__":
    main(sys.argv[1])
