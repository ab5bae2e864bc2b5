#!/bin/bash
# adaptive split parts: tests first, then the three scenes
O=gpurun_out/r06f; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_binsort.py tests/test_gpu_window.py tests/test_gpu_refine.py tests/test_gpu_parity.py tests/test_gpu_lineage_spec.py -x -q 2>&1 | tail -6 > $O/pytest.txt; cat $O/pytest.txt
python tools/scene_lists.py 180 600000 300 > $O/replica.json 2>>$O/err.txt
python tools/scene_lists.py 60 200000 300 > $O/room.json 2>>$O/err.txt
python tools/refine_idle.py S2-ref-layout 300 > $O/uniform.json 2>>$O/err.txt
python - <<PY
import json
for n in ("replica","room"):
    j=json.load(open("$O/%s.json"%n)); print(n,"refine_us", j["refine_us_per_iteration"], [(k["kernel"][9:36],k["us"]) for k in j["kernels"][:5]])
j=json.load(open("$O/uniform.json")); print("uniform", j["wall_us_per_iteration"], [(k["kernel"][9:36],k["us"]) for k in j["kernel_table_us_per_iteration"][:4]])
PY
tail -2 $O/err.txt
