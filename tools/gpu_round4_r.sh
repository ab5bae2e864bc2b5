#!/bin/bash
R=$(pwd)
mkdir -p gpurun_out/r4r
cd /tmp && export TMPDIR=/tmp
for v in base f2bfix; do
  if [ $v = f2bfix ]; then export SPLATRASTER_LIB=$R/splatloc_amd/_lib/variants/libsplatraster_f2bfix.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${v}_$c -o p -- python3 $R/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-multi-stream > /dev/null 2>&1
  done
done
cd $R
python3 - <<'PY'
import glob, sqlite3
for v in ("base","f2bfix"):
    for c in ("FETCH_SIZE","WRITE_SIZE"):
        db=sorted(glob.glob(f"/tmp/pmc_{v}_{c}/**/*.db", recursive=True))[0]
        con=sqlite3.connect(db)
        rows=con.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
        agg={}
        for n,cn,val in rows:
            k=n.split("(")[0].replace("void sr::","")
            if "composite" in k: a=agg.setdefault(k,[0,0]); a[0]+=val; a[1]+=1
        for k,(s,n) in agg.items(): print(v, c, k[:40], "KiB/launch", round(s/n))
PY
