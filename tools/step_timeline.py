#!/usr/bin/env python3
"""One train step of a rocprofv3 --kernel-trace (+ --memory-copy-trace) CSV as a timeline: every kernel / copy between the
end of one `sparse_apply_k` and the end of the next, with its start relative to the step's start, its duration, the
queue it ran on and the idle time of the whole GPU before it.
usage: step_timeline.py <kernel_trace.csv> [<memory_copy_trace.csv>] [--step K] [--mark NAME]   (K: which step, counted
from the middle of the trace; default 0 — the timed region of bench.py, before its instrumented pass; NAME: the kernel whose
end closes a step, default sparse_apply_k — the row-sharded step launches that one twice: --mark dense_apply_k)"""
import csv, sys

args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and not sys.argv[i - 1].startswith("--")]
k = int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else 0
mark = sys.argv[sys.argv.index("--mark") + 1] if "--mark" in sys.argv else "sparse_apply_k"
ev = []
for r in csv.DictReader(open(args[0])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.split("(")[0][:70], r.get("Queue_Id", "?")))
if len(args) > 1:
    for r in csv.DictReader(open(args[1])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", ""), "copy"))
ev.sort()
marks = [i for i, e in enumerate(ev) if e[2].startswith(mark)]
m0 = marks[len(marks) // 2 + k]
m1 = marks[len(marks) // 2 + k + 1]
t0 = ev[m0][1]
print("step of %.3f ms" % ((ev[m1][1] - t0) / 1e6))
print("%9s %9s %8s  %-6s %s" % ("start us", "dur us", "idle us", "queue", "kernel"))
reach = t0
busy = {}
for s, e, n, q in ev[m0 + 1:m1 + 1]:
    idle = max(0, s - reach)
    print("%9.1f %9.1f %8.1f  %-6s %s" % ((s - t0) / 1e3, (e - s) / 1e3, idle / 1e3, q, n))
    reach = max(reach, e)
    busy[q] = busy.get(q, 0) + e - s
print("busy per queue (ms):", {q: round(v / 1e6, 3) for q, v in busy.items()})
