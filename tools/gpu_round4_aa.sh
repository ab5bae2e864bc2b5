#!/bin/bash
mkdir -p gpurun_out/r4aa
python -m pytest tests/test_gpu_parity.py tests/test_gpu_window.py tests/test_gpu_edge_cases.py tests/test_gpu_refine.py -x -q 2>&1 | tail -4
python tools/ab.py --no-parity base f2bfix base f2bfix > gpurun_out/r4aa/ab_S2.txt 2>&1; cat gpurun_out/r4aa/ab_S2.txt
python tools/ab.py --no-parity --workload S2-ref-layout base f2bfix base f2bfix > gpurun_out/r4aa/ab_ref.txt 2>&1; cat gpurun_out/r4aa/ab_ref.txt
