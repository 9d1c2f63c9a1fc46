#!/usr/bin/env python3
"""Every number DESIGN.md 5 / BASELINE.md 4e quote from a closing run directory (gpurun_out/<tag> of tools/r06_final.sh), on a few lines.
usage: python tools/closing_numbers.py gpurun_out/r06v"""
import json
import os
import sys

X = sys.argv[1].rstrip("/") + "/"


def J(n):
    try:
        return json.load(open(X + n))
    except Exception as e:  # noqa: BLE001
        return {"_error": str(e)}


d = J("bench_default.json")
if "_error" not in d:
    r, s, c = d["roofline"], d["stream"], d["cpu_baseline"]
    print(f"default: value {d['value'] / 1e6:.2f} M/s, {d['ms_per_step']:.1f} ms/step, kernel {r['kernel_ms']:.1f} ms (HIP events), frac {r['frac']:.5f}, achieved {r['achieved']:.1f} GB/s, "
          f"traffic attached {r.get('traffic') is not None} {r.get('traffic_refused', '')}; longest query {d['tail_bound']['longest_query_ms']:.0f} ms")
    print(f"  stream {s['value'] / 1e6:.1f} M/s whole ({s['wall_s']:.1f} s), steady {s['steady_state_ms_per_batch']:.1f} ms/batch = {d['expansions_per_step'] / s['steady_state_ms_per_batch'] / 1e3:.1f} M/s, "
          f"latency {s['batch_latency_ms']}, batches {s['batches']} + {s['parity']['warmup_tickets_checked']}, mismatches {s['parity']['mismatches_vs_blocking_step']}")
    print(f"  cpu {c['value'] / 1e6:.3f} M/s on {c['cores']} ({c['value_per_core']:.0f} per core), single {c['single_thread']['value']:.0f}; parity {d['parity_sample']}; speculation {d['speculation']}")
    for k in ("c2", "c3", "c5", "lpa", "plpa"):
        e = d.get(k, {})
        print(f"  {k}: value {e.get('value')}, ms {e.get('ms_per_step')}, leg {e.get('leg_seconds')} s, vs cpu {e.get('vs_cpu_single_thread')}, parity {(e.get('parity_sample') or {}).get('mismatches')}, "
              f"frac {e.get('roofline', {}).get('frac')}, error {e.get('error')}")
print("wall:", open(X + "bench_default.wall").read().strip() if os.path.exists(X + "bench_default.wall") else "?")
t = J("traffic.json")
if "_error" not in t:
    kp = t["kernel_ms_counter_passes"]
    print(f"traffic: trace {t['kernel_ms_trace']:.1f} ms ({t['launches_in_trace']} launches), counter passes {min(kp):.0f}-{max(kp):.0f} ms ({len(kp)}), FETCH {t['FETCH_SIZE_KB'] / 1e6:.1f}e6 KB, "
          f"WRITE {t['WRITE_SIZE_KB'] / 1e6:.1f}e6 KB = {(t['FETCH_SIZE_KB'] + t['WRITE_SIZE_KB']) * 1024 / 1e9:.1f} GB, VALU {t['SQ_INSTS_VALU'] / 1e9:.1f} G, SALU {t['SQ_INSTS_SALU'] / 1e9:.1f} G, "
          f"wait {t['SQ_WAIT_ANY_over_WAVE_CYCLES']:.3f}, L2 hits {t['TCC_HIT_sum'] / 1e9:.2f} G misses {t['TCC_MISS_sum'] / 1e9:.2f} G; bulk kernel ms {t['bulk_phase']['kernel_ms'][:3]}")
for n in ("bench_c3", "bench_c4jrk", "bench_c5", "bench_lpa", "bench_plpa"):
    e = J(n + ".json")
    if "_error" in e:
        print(n, "missing:", e["_error"])
        continue
    cb = e.get("cpu_baseline") or {}
    print(f"{n}: value {e['value']}, {e['ms_per_step']:.1f} ms/step, cpu {cb.get('value')} ({cb.get('tick_ms', '')}), parity {e.get('parity_sample')}")
    if n == "bench_lpa":
        for r in e["cycle"]:
            print(f"    {r['step']}: lpa {r['lpa_ms']:.2f} ms, fresh {r['fresh_ms']:.2f} ms, update {r.get('update_ms', 0):.2f} ms, expansions {r['lpa_expansions']} / {r['fresh_expansions']}")
    if n == "bench_plpa":
        print("    lpa kernel ms", [round(r["lpa_kernel_ms"], 3) for r in e["replans"]], "fresh", [round(r["fresh_kernel_ms"], 3) for r in e["replans"]], "fresh wall mean", e["fresh_plan_wall_ms_mean"])
e = J("bench_c4jrk_2m.json")
if "_error" not in e:
    print(f"c4jrk_2m: {e['value'] / 1e6:.2f} M/s, {e['seconds']} s, {e['config']['launches']} launches of {e['config']['queries_per_launch']}, slots {e['config']['leading_workgroups']}, pools {e['config']['pools']}, "
          f"expansions {e['expansions']}, states {e['states_created']}, status {e['plan_status_counts']}, parity {e['parity_sample']['mismatches']} of {len(e['parity_sample']['queries'])}, "
          f"single {e['parity_sample']['same_words_from_single_plans_without_recycling']}, cpu per core {e['parity_sample']['cpu_expansions_per_s_per_core']:.0f}")
for n in ("full_parity_c4.json", "full_parity_c4jrk.json"):
    e = J(n)
    print(n, {k: e.get(k) for k in ("queries_replayed_on_cpu", "mismatches", "expansions", "cpu_expansions_per_s")})
for n in ("pytest_gpu.txt", "phase_tail.txt", "phase_bulk.txt"):
    if os.path.exists(X + n):
        L = open(X + n).read().strip().splitlines()
        print(n, "|", L[0][:120] if n.startswith("phase") else "", "|", L[-1][:120])
