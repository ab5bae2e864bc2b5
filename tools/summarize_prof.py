#!/usr/bin/env python3
"""Condenses rocprofv3 output (kernel-trace stats + PMC passes) into small text/JSON
summaries that are committed under profiles/.

usage: summarize_prof.py <rocprof_out_dir> <out_prefix>
Looks for *kernel_stats.csv (from --kernel-trace --stats) and *counter_collection.csv
(from --pmc FETCH_SIZE / --pmc WRITE_SIZE passes) anywhere below <rocprof_out_dir>.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    name = name.replace("void sr::", "").replace("sr::", "")
    return name.strip()


def main():
    root, prefix = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
    lines = []
    for path in sorted(glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)):
        lines.append(f"# {os.path.relpath(path, root)}")
        with open(path) as fh:
            rows = list(csv.DictReader(fh))
        lines.append(f"{'kernel':60s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
        for r in rows:
            lines.append(f"{short(r['Name'])[:60]:60s} {r['Calls']:>7s} {float(r['TotalDurationNs']) / 1e3:12.1f} "
                         f"{float(r['AverageNs']) / 1e3:10.2f} {float(r['MinNs']) / 1e3:10.2f} "
                         f"{float(r['MaxNs']) / 1e3:10.2f} {float(r['Percentage']):6.2f}")
    if lines:
        open(prefix + "_kernel_stats.txt", "w").write("\n".join(lines) + "\n")
    # PMC: per kernel sum / launches
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                k = short(r["Kernel_Name"])
                c = r["Counter_Name"]
                agg[k][c][0] += float(r["Counter_Value"])
                agg[k][c][1] += 1
    if agg:
        out = {}
        for k, cs in agg.items():
            out[k] = {c: {"sum": v[0], "dispatches": v[1], "per_dispatch": v[0] / max(v[1], 1)} for c, v in cs.items()}
        json.dump(out, open(prefix + "_pmc.json", "w"), indent=1, sort_keys=True)
    print("wrote", prefix + "_*")


if __name__ == "__main__":
    main()
