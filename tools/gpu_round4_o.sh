#!/bin/bash
mkdir -p gpurun_out/r4o
O=gpurun_out/r4o
python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_window.py tests/test_gpu_refine.py -x -q 2>&1 | tail -6 > $O/pytest_a.txt; cat $O/pytest_a.txt
python tools/grad_bar_probe.py > $O/grad_bars.txt 2>&1; cp gpurun_out/r4_grad_bars.json $O/
grep -v Warn $O/grad_bars.txt | grep -A7 "S2-ref-layout atomic mode 0 rtol 0.0001" | cut -c1-330
python tools/ab.py --no-parity --workload S2-ref-layout base f2bfix base f2bfix > $O/ab_ref.txt 2>&1; cat $O/ab_ref.txt
python tools/refine_idle.py S2-ref-layout 300 > $O/refine_idle.json 2>/dev/null; cut -c1-330 $O/refine_idle.json
