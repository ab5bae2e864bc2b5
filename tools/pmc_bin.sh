#!/bin/bash
# rocprofv3 SQ-counter passes over a native executable (tools/micro/bin/*); usage: tools/pmc_bin.sh OUTDIR FILTER exe [args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/$1; FILTER=$2; EXE=$R/$3; shift 3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
            "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES" \
            "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs -d $OUT/pass$i -o p -- $EXE "$@" > /dev/null 2> $OUT/pass$i.err || tail -3 $OUT/pass$i.err
done
cd $R
python3 - "$OUT" "$FILTER" <<'PY'
import glob, sqlite3, sys, json
from collections import defaultdict
out, flt = sys.argv[1], sys.argv[2]
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for db in glob.glob(out + "/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    for name, c, v in con.execute("select kernel_name, counter_name, value from counters_collection"):
        k = name.split("(")[0].replace("void sr::", "").replace("sr::", "")
        if any(f in k for f in flt.split(",")):
            agg[k][c][0] += v; agg[k][c][1] += 1
res = {k: {c: s / n for c, (s, n) in cs.items()} for k, cs in agg.items()}
json.dump(res, open(out + "/pmc_summary.json", "w"), indent=1, sort_keys=True)
for k, cs in res.items():
    print("==", k)
    for c in sorted(cs): print(f"   {c:28s} {cs[c]:16.0f}")
PY
find $OUT -name "*.db" -delete
