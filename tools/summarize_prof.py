#!/usr/bin/env python3
"""Condenses rocprofv3 output (rocpd sqlite: --kernel-trace --stats, and separate
--pmc FETCH_SIZE / --pmc WRITE_SIZE passes) into small text/JSON summaries for profiles/.

usage: summarize_prof.py <rocprof_out_dir> <out_prefix> [steps_per_pmc_run]

HBM traffic per launch follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests of wide coalesced
streaming reads at 64 B, so the read side is DOUBLED (upper bound for narrow accesses);
WRITE_SIZE is taken as reported (uncalibrated per the guide).
"""
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    return name.replace("void sr::", "").replace("sr::", "").replace("void ", "").strip()


def main():
    root, prefix = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
    lines = []
    pmc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for path in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
        con = sqlite3.connect(path)
        cur = con.cursor()
        n_pmc = cur.execute("select count(*) from counters_collection").fetchone()[0]
        if n_pmc == 0:
            lines.append(f"# rocprofv3 --kernel-trace --stats : {os.path.relpath(path, root)}")
            lines.append(f"{'kernel':44s} {'calls':>6s} {'total_us':>11s} {'avg_us':>10s} {'pct':>6s} {'vgpr':>5s} {'sgpr':>5s} {'lds_B':>7s} {'grid':>9s}")
            meta = {}
            for name, vg, sg, lds, gx, gy, gz in cur.execute(
                    "select name, vgpr_count, sgpr_count, lds_size, grid_x, grid_y, grid_z from kernels group by name"):
                meta[name] = (vg, sg, lds, gx * gy * gz)
            for name, calls, total, avg, pct in cur.execute(
                    "select name, total_calls, total_duration, average, percentage from top_kernels"):
                vg, sg, lds, grid = meta.get(name, (0, 0, 0, 0))
                lines.append(f"{short(name)[:44]:44s} {calls:6d} {total:11.1f} {avg:10.2f} {pct:6.2f} {vg:5d} {sg:5d} {lds:7d} {grid:9d}")
        else:
            for name, cname, val in cur.execute("select kernel_name, counter_name, value from counters_collection"):
                pmc[short(name)][cname][0] += float(val)
                pmc[short(name)][cname][1] += 1
        con.close()
    if lines:
        open(prefix + "_kernel_stats.txt", "w").write("\n".join(lines) + "\n")
        print("\n".join(lines))
    if pmc:
        out = {}
        for k, cs in pmc.items():
            e = {}
            for c, (s, n) in cs.items():
                e[c + "_KiB_per_launch"] = s / max(n, 1)
                e[c + "_launches"] = n
            rd = e.get("FETCH_SIZE_KiB_per_launch")
            wr = e.get("WRITE_SIZE_KiB_per_launch")
            if rd is not None and wr is not None:
                e["hbm_bytes_per_launch"] = int(rd * 1024 * 2 + wr * 1024)
                e["note"] = "read = 2 x FETCH_SIZE (gfx950 correction), write = WRITE_SIZE as reported"
            out[k] = e
        json.dump(out, open(prefix + "_pmc.json", "w"), indent=1, sort_keys=True)
        for k in sorted(out, key=lambda k: -out[k].get("hbm_bytes_per_launch", 0))[:12]:
            print(f"{k[:44]:44s}", {a: (round(b, 1) if isinstance(b, float) else b) for a, b in out[k].items() if a != "note"})
    print("wrote", prefix + "_*")


if __name__ == "__main__":
    main()
