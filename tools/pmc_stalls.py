#!/usr/bin/env python3
"""Memory-side request / stall counters of one kernel from rocprofv3 --pmc passes (TCC block: 4 counters per pass):
per dispatch medians, and ratios to the L2's busy cycles.
usage: pmc_stalls.py <kernel name substring> <counter_collection.csv> [<counter_collection.csv> ...]"""
import csv, statistics, sys
want = sys.argv[1]
vals = {}
for path in sys.argv[2:]:
    per = {}
    for r in csv.DictReader(open(path)):
        if want in r["Kernel_Name"]:
            per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in per.items():
        vals[k] = (statistics.median(v), len(v))
print("| counter | median per dispatch | dispatches |")
print("|---|---:|---:|")
for k in sorted(vals):
    print("| `%s` | %.4g | %d |" % (k, vals[k][0], vals[k][1]))
