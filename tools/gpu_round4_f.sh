#!/bin/bash
mkdir -p gpurun_out/r4f
O=gpurun_out/r4f
python tools/hostprof_steps.py refine S0 300 > $O/hostprof_refine.txt 2>&1; head -100 $O/hostprof_refine.txt
python tools/hostprof_steps.py map S0 150 > $O/hostprof_map.txt 2>&1; head -100 $O/hostprof_map.txt
python -m pytest tests/test_gpu_parity.py -x -q -k "dist2" 2>&1 | tail -4
