#!/bin/bash
mkdir -p gpurun_out/r4bb
python -m pytest tests/test_gpu_refine.py tests/test_gpu_training.py tests/test_gpu_scene.py tests/test_gpu_densify.py tests/test_gpu_keyframe.py -x -q 2>&1 | tail -6
python bench.py --stage map_step --workload S2-ref-layout --steps 100 --warmup 10 > gpurun_out/r4bb/map_ref.json 2>/dev/null; cut -c1-260 gpurun_out/r4bb/map_ref.json; echo
python bench.py --stage map_step > gpurun_out/r4bb/map_S2.json 2>/dev/null; cut -c1-260 gpurun_out/r4bb/map_S2.json; echo
python bench.py --stage scene > gpurun_out/r4bb/scene.json 2>/dev/null; python -c "
import json; j=json.load(open('gpurun_out/r4bb/scene.json')); print(j['seconds'], j['map_ms_per_iteration'], j['refine_ms_per_iteration'], j['eval'])"
