#!/usr/bin/env python3
"""Kernel timeline of one bench step from a rocprofv3 --kernel-trace rocpd database:
start offset, duration and the idle gap before every kernel, plus busy/idle totals.
usage: timeline.py <dir with *.db> [step index from the end, default 2]"""
import glob
import sqlite3
import sys


def main():
    root = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    db = sorted(glob.glob(root + "/**/*.db", recursive=True))[0]
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    # a step starts at every preprocess_kernel
    starts = [i for i, r in enumerate(rows) if "preprocess_kernel" in r[0] and "bwd" not in r[0]]
    a = starts[-back - 1]
    b = starts[-back]
    step = rows[a:b]
    t0 = step[0][1]
    busy = 0
    prev_end = t0
    print(f"{'kernel':40s} {'start_us':>9s} {'dur_us':>8s} {'gap_us':>7s}")
    for name, s, e in step:
        n = name.split("(")[0].replace("void sr::", "").replace("sr::", "").replace("void ", "")[:40]
        print(f"{n:40s} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f}")
        busy += e - s
        prev_end = max(prev_end, e)
    nxt = rows[b][1]
    print(f"step span {(nxt - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, idle {(nxt - t0 - busy) / 1e3:.1f} us, "
          f"tail gap to next step {(nxt - prev_end) / 1e3:.1f} us, {len(step)} kernels")


if __name__ == "__main__":
    main()
