#!/bin/bash
mkdir -p gpurun_out/r4d
O=gpurun_out/r4d
python -m pytest tests/test_gpu_eval.py tests/test_gpu_keyframe.py -x -q 2>&1 | tail -12 > $O/pytest_eval.txt; cat $O/pytest_eval.txt
python -m pytest tests/test_gpu_scene.py -x -q 2>&1 | tail -25 > $O/pytest_scene.txt; cat $O/pytest_scene.txt
python -m pytest tests/test_gpu_window.py tests/test_gpu_lineage_spec.py -x -q 2>&1 | tail -12 > $O/pytest_win.txt; cat $O/pytest_win.txt
python bench.py --stage eval_rendering --steps 5 --warmup 1 > $O/eval.json 2> $O/eval.err; tail -2 $O/eval.err; cut -c1-600 $O/eval.json
python bench.py --stage scene > $O/scene.json 2> $O/scene.err; tail -3 $O/scene.err; cut -c1-1500 $O/scene.json
