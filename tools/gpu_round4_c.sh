#!/bin/bash
mkdir -p gpurun_out/r4c
O=gpurun_out/r4c
python -m pytest tests/test_gpu_parity.py tests/test_gpu_window.py -x -q 2>&1 | tail -8 > $O/pytest_a.txt; cat $O/pytest_a.txt
python tools/grad_bar_probe.py > $O/grad_bars.txt 2>&1; grep -v Warn $O/grad_bars.txt | grep -A7 "det mode . rtol 0.0001" | cut -c1-330
cp gpurun_out/r4_grad_bars.json $O/ 2>/dev/null
python bench.py --no-cpu-baseline --no-multi-stream > $O/bench_S2.json 2> $O/bench_S2.err; cut -c1-300 $O/bench_S2.json
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py --deselect tests/test_gpu_window.py 2>&1 | tail -6 > $O/pytest_b.txt; cat $O/pytest_b.txt
