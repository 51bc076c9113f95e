#!/usr/bin/env python3
"""Matrix-pipe utilisation of the MLP GEMM kernels from one rocprofv3 --pmc pass:
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d <dir> -o run -- python3 bench.py ...
busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) per dispatch: the first counter sums the cycles
each of the chip's 1024 matrix pipes is busy; GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (a 317 us launch reads
4.3 M = 8 x 539 k cycles, i.e. 1.7 GHz under matrix load), so one eighth of it is the dispatch's duration in shader
clocks.  Median over the dispatches.
usage: mfma_busy.py <counter_collection.csv> > profiles/rNN_mfma_busy.md"""
import csv, re, statistics, sys

per = {}
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if "anonymous namespace" not in name or "at::native" in name:
        continue
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"^void ", "", short).split("(")[0]
    per.setdefault((short, r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
agg = {}
for (k, _), c in per.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE", 0) > 0 and c["SQ_VALU_MFMA_BUSY_CYCLES"] > 0:
        agg.setdefault(k, []).append((c["SQ_VALU_MFMA_BUSY_CYCLES"], c["GRBM_GUI_ACTIVE"]))
print("| kernel | dispatches | SQ_VALU_MFMA_BUSY_CYCLES (median) | GRBM_GUI_ACTIVE / 8 (median) | matrix pipes busy |")
print("|---|---:|---:|---:|---:|")
for k, v in sorted(agg.items(), key=lambda kv: -statistics.median(x[0] for x in kv[1])):
    b = statistics.median(x[0] for x in v); g = statistics.median(x[1] for x in v)
    print("| `%s` | %d | %.3g | %.3g | %.1f %% |" % (k, len(v), b, g / 8, 100.0 * b / (g / 8 * 1024)))
