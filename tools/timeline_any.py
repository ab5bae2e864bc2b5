#!/usr/bin/env python3
"""Kernel timeline of the LAST `n` kernels before the end of a rocprofv3 --kernel-trace run, grouped into steps that start at
every `marker` kernel: start offset, duration, idle gap.  usage: timeline_any.py <dir with *.db> <marker substring> [step index from the end]"""
import glob
import sqlite3
import sys

root, marker = sys.argv[1], sys.argv[2]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
db = sorted(glob.glob(root + "/**/*.db", recursive=True))[0]
rows = sqlite3.connect(db).execute("select name, start, end from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if marker in r[0]]
a, b = starts[-back - 1], starts[-back]
step = rows[a:b]
t0 = step[0][1]
busy, prev_end = 0, t0
print(f"{'kernel':44s} {'start_us':>9s} {'dur_us':>8s} {'gap_us':>7s}")
for name, s, e in step:
    n = name.split("(")[0].replace("void sr::", "").replace("sr::", "").replace("void ", "")[:44]
    print(f"{n:44s} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f}")
    busy += e - s
    prev_end = max(prev_end, e)
nxt = rows[b][1]
print(f"step span {(nxt - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, idle {(nxt - t0 - busy) / 1e3:.1f} us, {len(step)} kernels")
