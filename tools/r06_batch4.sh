#!/bin/bash
# round-6 batch: full GPU suite, map-step idle (frozen scene), Replica-scale refinement iteration under the sort-launch modes
O=gpurun_out/r06d; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest.txt; cat $O/pytest.txt
python tools/map_idle.py S2-ref-layout 100 0 > $O/map_idle_ref.json 2>$O/err.txt; cut -c1-420 $O/map_idle_ref.json; echo
python tools/map_idle.py S2 40 0 > $O/map_idle_S2.json 2>>$O/err.txt; cut -c1-420 $O/map_idle_S2.json; echo
for mode in "0 0" "-1 0" "1 2048" "1 4096"; do set -- $mode
  SPLATRASTER_SORT_FORK=$1 SPLATRASTER_TILE_SORT_CAP=$2 python tools/scene_lists.py 180 600000 300 > $O/scene_lists_replica_fork$1_cap$2.json 2>>$O/err.txt
  python - <<PY
import json; j=json.load(open("$O/scene_lists_replica_fork$1_cap$2.json"))
print("replica-scale fork=$1 cap=$2 refine_us", j["refine_us_per_iteration"], [(k["kernel"][9:40],k["us"]) for k in j["kernels"][:5]])
PY
done
for f in 0 1; do SPLATRASTER_SORT_FORK=$f python tools/refine_idle.py S2-ref-layout 300 > $O/refine_idle_fork$f.json 2>>$O/err.txt; cut -c1-330 $O/refine_idle_fork$f.json; echo; done
for f in 0 1; do SPLATRASTER_SORT_FORK=$f python tools/scene_lists.py 60 200000 300 > $O/scene_lists_room_fork$f.json 2>>$O/err.txt
  python - <<PY
import json; j=json.load(open("$O/scene_lists_room_fork$f.json"))
print("room fork=$f refine_us", j["refine_us_per_iteration"], [(k["kernel"][9:40],k["us"]) for k in j["kernels"][:5]])
PY
done
python bench.py --no-cpu-baseline --no-multi-stream --workload S2-ref-layout > $O/bench_ref.json 2>>$O/err.txt; cut -c1-200 $O/bench_ref.json; echo
tail -3 $O/err.txt
