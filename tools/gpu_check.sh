#!/bin/bash
# On the GPU box: window tests first (fast fail), the rest of the GPU suite, then the bench in both layouts.
# usage: tools/gpu_check.sh TAG
TAG=${1:-chk}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_window.py tests/test_gpu_keyframe.py tests/test_gpu_refine.py -x -q 2>&1 | tail -25 > gpurun_out/${TAG}_window.txt; cat gpurun_out/${TAG}_window.txt
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_window.py --deselect tests/test_gpu_keyframe.py --deselect tests/test_gpu_refine.py 2>&1 | tail -8 > gpurun_out/${TAG}_pytest.txt; cat gpurun_out/${TAG}_pytest.txt
for wl in S2 S2-ref-layout; do
  python bench.py --no-cpu-baseline --workload $wl > gpurun_out/${TAG}_bench_$wl.json 2> gpurun_out/${TAG}_bench_$wl.err; tail -3 gpurun_out/${TAG}_bench_$wl.err
  python - <<PY
import json
j=json.load(open("gpurun_out/${TAG}_bench_$wl.json"))
print("$wl", j["value"], j["ms_per_step"], j["roofline"]["avg_ms"], j["roofline"]["frac"], j["multi_stream"])
print({k:(v["avg_ms"],v["launches"]) for k,v in j["stages"].items()})
PY
done
for st in map_step refine_step; do
  python bench.py --stage $st --workload S2-ref-layout > gpurun_out/${TAG}_stage_${st}_ref.json 2> gpurun_out/${TAG}_stage_${st}_ref.err; tail -2 gpurun_out/${TAG}_stage_${st}_ref.err; cut -c1-900 gpurun_out/${TAG}_stage_${st}_ref.json
done
python bench.py --stage map_step > gpurun_out/${TAG}_stage_map_step_S2.json 2> gpurun_out/${TAG}_stage_map_step_S2.err; cut -c1-600 gpurun_out/${TAG}_stage_map_step_S2.json
