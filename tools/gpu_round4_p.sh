#!/bin/bash
mkdir -p gpurun_out/r4p
O=gpurun_out/r4p
for thr in 26000 6144 0; do
  echo "== split max waves $thr"
  SPLATRASTER_SPLIT_MAX_WAVES=$thr python bench.py --no-cpu-baseline --no-multi-stream --workload S2-ref-layout --steps 30 > $O/ref_$thr.json 2>/dev/null
  python - <<PY
import json
j=json.load(open("$O/ref_$thr.json")); print("window5", j["value"], {k:v["avg_ms"] for k,v in j["stages"].items() if "composite" in k})
PY
  SPLATRASTER_SPLIT_MAX_WAVES=$thr python tools/refine_idle.py S2-ref-layout 300 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('refine wall', j['wall_us_per_iteration'], 'busy', j['gpu_busy_us_per_iteration_torch_profiler'], [(r['kernel'][:34], r['us']) for r in j['kernel_table_us_per_iteration'][:2]])"
done
