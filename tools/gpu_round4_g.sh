#!/bin/bash
mkdir -p gpurun_out/r4g
O=gpurun_out/r4g
python -m pytest tests/test_gpu_refine.py tests/test_gpu_scene.py tests/test_gpu_training.py tests/test_gpu_densify.py -x -q 2>&1 | tail -8 > $O/pytest.txt; cat $O/pytest.txt
python tools/refine_idle.py S2-ref-layout 300 > $O/refine_idle.json 2> /dev/null; cut -c1-420 $O/refine_idle.json; echo
python tools/refine_idle.py S0 300 > $O/refine_idle_S0.json 2> /dev/null; cut -c1-420 $O/refine_idle_S0.json; echo
python bench.py --stage scene > $O/scene.json 2> $O/scene.err; tail -2 $O/scene.err; cut -c1-1200 $O/scene.json; echo
python bench.py --stage refine_step --workload S2-ref-layout --steps 300 --warmup 30 > $O/refine.json 2>/dev/null; cut -c1-300 $O/refine.json; echo
python bench.py --stage map_step --workload S2-ref-layout --steps 100 --warmup 10 > $O/map.json 2>/dev/null; cut -c1-300 $O/map.json; echo
