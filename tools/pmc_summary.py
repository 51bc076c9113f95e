#!/usr/bin/env python3
"""Condenses two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: one counter per pass, as the MI355X guide
prescribes) into a per-kernel table of HBM bytes per launch and refreshes profiles/traffic.json.
gfx950 correction: FETCH_SIZE reports half the bytes of wide coalesced reads -> read bytes = 2 x FETCH_SIZE;
WRITE_SIZE is exact; unit KiB per dispatch.  Median over the dispatches of the run.
usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> [traffic.json] > profiles/rNN_pmc.md"""
import csv, json, re, statistics, sys


def load(path, counter):
    per = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        if "anonymous namespace" not in name or "at::native" in name:
            continue
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"^void ", "", short).split("(")[0]
        per.setdefault(short, []).append(float(r["Counter_Value"]))
    return {k: statistics.median(v) for k, v in per.items()}, {k: len(v) for k, v in per.items()}


fetch, nf = load(sys.argv[1], "FETCH_SIZE")
write, _ = load(sys.argv[2], "WRITE_SIZE")
rows = []
for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0))):
    rd, wr = 2 * fetch[k] * 1024, write.get(k, 0) * 1024
    rows.append((k, nf[k], fetch[k], rd, write.get(k, 0), wr))
print("| kernel | dispatches | FETCH_SIZE raw (KiB) | x2 read MB | WRITE_SIZE (KiB) | write MB | HBM MB per launch |")
print("|---|---:|---:|---:|---:|---:|---:|")
for k, n, f, rd, w, wr in rows:
    print("| `%s` | %d | %.0f | %.1f | %.0f | %.1f | %.1f |" % (k, n, f, rd / 1e6, w, wr / 1e6, (rd + wr) / 1e6))
if len(sys.argv) > 3:
    out = {"_comment": "HBM bytes per launch from rocprofv3 PMC passes (2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes), tools/pmc_summary.py; "
                       "bench.py copies the entry of the roofline kernel into roofline.traffic"}
    for k, n, f, rd, w, wr in rows:
        base = k.split("<")[0]
        if base in ("embed_fm_planes_fwd_k", "embed_fm_linear_fwd_k", "sparse_apply_k", "sparse_catchup_k", "sparse_catchup_bounded_k",
                    "catchup_lin_k") and base not in out:
            out[base] = {"bytes_per_launch": int(rd + wr), "fetch_size_kib_raw": int(f), "write_size_kib": int(w)}
    json.dump(out, open(sys.argv[3], "w"), indent=2)
