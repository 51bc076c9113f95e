#!/usr/bin/env python3
"""Condenses a rocprofv3 --kernel-trace --stats CSV (kernel_stats.csv) into a short markdown table:
this repo's kernels by name, everything else (torch init / RNG / copies) lumped together.
usage: prof_summary.py <kernel_stats.csv> [<bench json line file>] > profiles/rNN_xxx.md"""
import csv, json, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
ours, other_ns, other_calls = [], 0, 0
for r in rows:
    name = r["Name"]
    if "anonymous namespace" in name and "at::native" not in name:
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"^void ", "", short).split("(")[0]
        ours.append((short, int(r["Calls"]), int(r["TotalDurationNs"]), float(r["AverageNs"]), int(r["MinNs"]), int(r["MaxNs"])))
    else:
        other_ns += int(r["TotalDurationNs"]); other_calls += int(r["Calls"])
tot = sum(o[2] for o in ours)
print("| kernel | calls | total ms | avg us | min us | max us | % of our kernels |")
print("|---|---:|---:|---:|---:|---:|---:|")
for o in sorted(ours, key=lambda x: -x[2]):
    print("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.1f |" % (o[0], o[1], o[2] / 1e6, o[3] / 1e3, o[4] / 1e3, o[5] / 1e3, 100.0 * o[2] / tot))
print("| torch init / RNG / fills (not in the timed region) | %d | %.3f | | | | |" % (other_calls, other_ns / 1e6))
if len(sys.argv) > 2:
    for l in open(sys.argv[2]):
        if l.startswith("{"):
            d = json.loads(l)
            print("\nbench line of the same run: value %.0f %s, %.3f ms/step, gather %.0f GB/s (avg launch %.1f us), MLP %.1f TFLOP/s" % (
                d["value"], d["unit"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["avg_launch_ms"] * 1e3, d["roofline_mlp"]["achieved"]))
