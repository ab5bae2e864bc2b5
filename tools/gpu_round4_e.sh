#!/bin/bash
mkdir -p gpurun_out/r4e
O=gpurun_out/r4e
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest.txt; cat $O/pytest.txt
python tools/refine_idle.py S2-ref-layout 300 > $O/refine_idle.json 2> $O/refine_idle.err; tail -2 $O/refine_idle.err; cut -c1-700 $O/refine_idle.json
python tools/refine_idle.py S0 300 > $O/refine_idle_S0.json 2> /dev/null; cut -c1-400 $O/refine_idle_S0.json
