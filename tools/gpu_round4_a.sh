#!/bin/bash
# round 4, first GPU call: new multi-rank tests, the whole GPU suite, baseline bench, what the box exposes for clocks, clock trace
mkdir -p gpurun_out/r4a
O=gpurun_out/r4a
python -m pytest tests/test_gpu_training.py -x -q 2>&1 | tail -15 > $O/training.txt; cat $O/training.txt
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_training.py 2>&1 | tail -8 > $O/pytest.txt; cat $O/pytest.txt
python bench.py --no-cpu-baseline > $O/bench_S2.json 2> $O/bench_S2.err; tail -2 $O/bench_S2.err; cut -c1-400 $O/bench_S2.json
( ls /sys/class/drm/; ls /sys/class/drm/card*/device/ | head -80; ls /sys/class/drm/card*/device/hwmon/*/; which amd-smi rocm-smi; (time amd-smi metric -g 0 --clock --power --usage --json) 2>&1 | head -120 ) > $O/probe.txt 2>&1
python tools/clock_trace.py $O/clocks.json > $O/clock_summary.txt 2>&1; cat $O/clock_summary.txt
