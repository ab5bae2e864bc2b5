#!/bin/bash
mkdir -p gpurun_out/r4q
O=gpurun_out/r4q
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt; cat $O/pytest.txt
python bench.py --no-cpu-baseline > $O/bench_S2.json 2>/dev/null; cut -c1-260 $O/bench_S2.json; echo
python bench.py --no-cpu-baseline --workload S2-ref-layout > $O/bench_ref.json 2>/dev/null; cut -c1-260 $O/bench_ref.json; echo
