#!/bin/bash
mkdir -p gpurun_out/r4i
O=gpurun_out/r4i
python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -12 > $O/pytest_a.txt; cat $O/pytest_a.txt
python -m pytest tests/test_gpu_window.py tests/test_gpu_lineage_spec.py tests/test_gpu_refine.py tests/test_gpu_pose.py -x -q 2>&1 | tail -12 > $O/pytest_b.txt; cat $O/pytest_b.txt
python tools/ab.py --no-parity base f2b base f2b > $O/ab_S2.txt 2>&1; cat $O/ab_S2.txt
python tools/ab.py --no-parity --workload S2-ref-layout base f2b base f2b > $O/ab_ref.txt 2>&1; cat $O/ab_ref.txt
