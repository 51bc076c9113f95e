#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace (+ --memory-copy-trace) CSV pair: the last `nsteps` train steps' GPU timeline —
busy time per kernel name (all kernels, ours and torch's / RCCL's), copies, and idle time between them.
usage: trace_gaps.py <kernel_trace.csv> [<memory_copy_trace.csv>] [--window-ms 40] [--end-offset-ms 0]
(bench.py ends with a pass that brackets every launch with HIP events — ~10 us of bubble per launch; --end-offset-ms
moves the window's end back past it, into the timed region.)"""
import csv, sys, collections

win_ms = 40.0
args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and not sys.argv[i - 1].startswith("--")]
if "--window-ms" in sys.argv:
    win_ms = float(sys.argv[sys.argv.index("--window-ms") + 1])
ev = []
for r in csv.DictReader(open(args[0])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.split("(")[0][:90]))
if len(args) > 1:
    for r in csv.DictReader(open(args[1])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ev.sort()
# the window: the last win_ms of the trace that contains our sharded-step kernels
marks = [e for e in ev if "sparse_apply_k" in e[2]]
t_end = marks[-1][1]
if "--end-offset-ms" in sys.argv:
    t_end -= int(float(sys.argv[sys.argv.index("--end-offset-ms") + 1]) * 1e6)
t_beg = t_end - int(win_ms * 1e6)
sel = [e for e in ev if e[0] >= t_beg and e[1] <= t_end]
busy = collections.Counter(); calls = collections.Counter()
cover = []
for s, e, n in sel:
    busy[n] += e - s; calls[n] += 1
    cover.append((s, e))
cover.sort()
tot_busy, cur_s, cur_e = 0, None, None
for s, e in cover:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            tot_busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
tot_busy += cur_e - cur_s
span = t_end - t_beg
print("window %.1f ms: GPU busy %.2f ms (%.0f %%), idle %.2f ms" % (span / 1e6, tot_busy / 1e6, 100 * tot_busy / span, (span - tot_busy) / 1e6))
# the largest idle gaps of the window and what surrounds them
sel.sort()
gaps, reach, last = [], None, None
for s0, e0, n0 in sel:
    if reach is not None and s0 > reach:
        gaps.append((s0 - reach, last, n0))
    if reach is None or e0 > reach:
        reach, last = e0, n0
print("largest gaps (us): after -> before")
for g, a, b in sorted(gaps, reverse=True)[:14]:
    print("%8.1f  %s -> %s" % (g / 1e3, a[:50], b[:50]))
agg = collections.Counter()
for g, a, b in gaps:
    agg[(a[:40], b[:40])] += g
print("gap time by (kernel before -> after), ms:")
for (a, b), g in agg.most_common(12):
    print("%8.3f  %s -> %s" % (g / 1e6, a, b))
for n, b in busy.most_common(25):
    print("%9.3f ms %6d x  %s" % (b / 1e6, calls[n], n))
