#!/bin/bash
mkdir -p gpurun_out/r4l
bash tools/pmc_passes.sh gpurun_out/r4l/sq_b2f > gpurun_out/r4l/new.txt 2>&1
export SPLATRASTER_LIB=$(pwd)/splatloc_amd/_lib/variants/libsplatraster_f2bfix.so
bash tools/pmc_passes.sh gpurun_out/r4l/sq_f2bfix > gpurun_out/r4l/old.txt 2>&1
python - <<'PY'
import json
a=json.load(open('gpurun_out/r4l/sq_b2f/pmc_summary.json')); b=json.load(open('gpurun_out/r4l/sq_f2bfix/pmc_summary.json'))
ka=[k for k in a if 'composite_bwd' in k][0]; kb=[k for k in b if 'composite_bwd' in k][0]
for c in sorted(a[ka]):
    x,y=a[ka][c],b[kb].get(c,0)
    print(f"{c:30s} b2f {x:16.0f} f2bfix {y:16.0f} ratio {x/max(y,1):.3f}")
PY
