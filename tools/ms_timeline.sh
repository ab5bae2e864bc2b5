#!/bin/bash
# kernel timeline of one map_step / refine_step between two fused Adam launches, under rocprofv3: tools/ms_timeline.sh [workload] [stage]
WL=${1:-S2-ref-layout}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/ms -o ms -- python3 $R/bench.py --stage ${2:-map_step} --workload $WL > /tmp/ms.json 2> /tmp/ms.err
tail -3 /tmp/ms.err
cd $R
python3 - <<'PY' > gpurun_out/map_step_timeline.txt 2>&1
import glob, sqlite3
db = sorted(glob.glob("/tmp/ms/**/*.db", recursive=True))[0]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t == "kernels"] or [t for t in tabs if "kernel" in t.lower()]
rows = con.execute(f"select name, start, end from {kt[0]} order by start").fetchall()
names = [r[0].split("(")[0].replace("void sr::", "").replace("sr::", "").replace("void ", "") for r in rows]
# one step of OUR map_step = from one fused Adam launch to the next
idx = [i for i, n in enumerate(names) if n.startswith("adam_")]
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
t0 = rows[a + 1][1]; prev = t0; busy = 0
for (name, s, e), n in zip(rows[a + 1:b + 1], names[a + 1:b + 1]):
    print(f"{n[:50]:50s} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev) / 1e3:7.1f}")
    busy += e - s; prev = max(prev, e)
print(f"span {(rows[b + 1][1] - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, kernels {b - a}")
from collections import Counter
cnt = Counter(r[0][:230] for r in rows[a + 1:b + 1] if r[0].startswith("void at::") or r[0].startswith("at::"))
for k, v in cnt.most_common(30): print(v, k)
PY
tail -80 gpurun_out/map_step_timeline.txt
