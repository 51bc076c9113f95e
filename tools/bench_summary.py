#!/usr/bin/env python3
"""One line per bench.py log: ms/step, cold start, the sparse side's per-entry times, the extra legs.
usage: python tools/bench_summary.py gpurun_out/<log> [...]"""
import json
import sys

for path in sys.argv[1:]:
    for line in open(path):
        if not line.startswith("{"):
            continue
        d = json.loads(line)
        k = d["kernel_ms_per_step"]
        print(path, "ms/step %.3f" % d["ms_per_step"], "cold %.3f" % d["cold_start"]["ms_per_step"],
              {x: round(v, 3) for x, v in k.items() if any(t in x for t in ("sparse", "catchup", "mark", "sort"))},
              {kk: round(v["ms_per_step"], 3) for kk, v in d.items()
               if isinstance(v, dict) and "ms_per_step" in v and kk != "cold_start"})
