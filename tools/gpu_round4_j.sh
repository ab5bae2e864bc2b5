#!/bin/bash
mkdir -p gpurun_out/r4j
bash tools/pmc_passes.sh gpurun_out/r4j/sq_new > gpurun_out/r4j/new.txt 2>&1
export SPLATRASTER_LIB=$(pwd)/splatloc_amd/_lib/variants/libsplatraster_f2b.so
bash tools/pmc_passes.sh gpurun_out/r4j/sq_old > gpurun_out/r4j/old.txt 2>&1
python - <<'PY'
import json
a=json.load(open('gpurun_out/r4j/sq_new/pmc_summary.json')); b=json.load(open('gpurun_out/r4j/sq_old/pmc_summary.json'))
ka=[k for k in a if 'composite_bwd' in k][0]; kb=[k for k in b if 'composite_bwd' in k][0]
print(ka, kb)
for c in sorted(a[ka]):
    x,y=a[ka][c],b[kb].get(c,0)
    print(f"{c:30s} new {x:16.0f} old {y:16.0f} ratio {x/max(y,1):.3f}")
PY
