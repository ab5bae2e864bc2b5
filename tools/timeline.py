#!/usr/bin/env python3
"""Kernel timeline of ONE TIMED bench step from a rocprofv3 --kernel-trace rocpd database: start offset, duration and the
idle gap before every kernel, plus busy / idle totals, reconciled with the bench line's ms_per_step.

usage: timeline.py <dir with *.db> [bench JSON written by the same run]

A step starts at every preprocess_kernel launch.  bench.py runs, in this order: `warmup` steps, one recording step,
BREAKDOWN_STEPS = 20 steps with EVERY stage bracketed by HIP events (an event record idles the GPU ~10 us: those steps are
longer and are not what `value` measures), then `repeats` timed regions of `steps` steps (only the dominant kernel bracketed),
then — unless --no-multi-stream — secondary legs.  The step printed is the middle step of the middle timed region, located
from the counts in the bench JSON (round 4 printed "the 2nd step from the end", which was a secondary-leg step: its span did
not reconcile with ms_per_step).  Without a JSON the counts of tools/profile_round.sh are assumed (3 warm-up, 10 steps, 5 regions)."""
import glob
import json
import sqlite3
import sys

BREAKDOWN_STEPS = 20


def main():
    root = sys.argv[1]
    meta = {"warmup": 3, "steps": 10, "repeats": {"regions": 5}}
    if len(sys.argv) > 2:
        with open(sys.argv[2]) as f:
            line = [l for l in f.read().splitlines() if l.startswith("{")][-1]
        meta = json.loads(line)
    warmup, steps = int(meta["warmup"]), int(meta["steps"])
    regions = int(meta.get("repeats", {}).get("regions", 5))
    db = sorted(glob.glob(root + "/**/*.db", recursive=True))[0]
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    starts = [i for i, r in enumerate(rows) if "preprocess_kernel" in r[0] and "bwd" not in r[0]]
    first_timed = warmup + 1 + BREAKDOWN_STEPS
    k = first_timed + steps * (regions // 2) + steps // 2
    if k + 1 >= len(starts):
        raise SystemExit(f"timeline.py: {len(starts)} steps in the trace, the timed step {k} is not among them (wrong bench JSON?)")
    a, b = starts[k], starts[k + 1]
    step = rows[a:b]
    t0 = step[0][1]
    busy = 0
    prev_end = t0
    print(f"# step {k} of {len(starts)} in the trace = step {steps // 2} of timed region {regions // 2} "
          f"(warm-up {warmup}, 1 recording, {BREAKDOWN_STEPS} event-bracketed, then {regions} x {steps} timed)")
    print(f"{'kernel':40s} {'start_us':>9s} {'dur_us':>8s} {'gap_us':>7s}")
    for name, s, e in step:
        n = name.split("(")[0].replace("void sr::", "").replace("sr::", "").replace("void ", "")[:40]
        print(f"{n:40s} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f}")
        busy += e - s
        prev_end = max(prev_end, e)
    nxt = rows[b][1]
    span = (nxt - t0) / 1e3
    msg = (f"step span {span:.1f} us, kernels busy {busy / 1e3:.1f} us, idle {(nxt - t0 - busy) / 1e3:.1f} us, "
           f"tail gap to next step {(nxt - prev_end) / 1e3:.1f} us, {len(step)} kernels")
    if "ms_per_step" in meta:
        msg += f"; the bench line of this run: ms_per_step {meta['ms_per_step']} ({100.0 * span / (1e3 * meta['ms_per_step']) - 100.0:+.1f} % off)"
    print(msg)


if __name__ == "__main__":
    main()
