#!/bin/bash
O=gpurun_out/r06g; mkdir -p $O
V=$PWD/splatloc_amd/_lib/variants
for v in k256 k256l15 k512l1 base; do
  if [ $v = base ]; then unset SPLATRASTER_LIB; else export SPLATRASTER_LIB=$V/libsplatraster_$v.so; fi
  python tools/scene_lists.py 180 600000 300 > $O/replica_$v.json 2>>$O/err.txt
  python tools/scene_lists.py 60 200000 300 > $O/room_$v.json 2>>$O/err.txt
  python tools/refine_idle.py S2-ref-layout 300 > $O/uniform_$v.json 2>>$O/err.txt
  python - <<PY
import json
for n in ("replica","room"):
    j=json.load(open("$O/%s_$v.json"%n)); print("$v",n,"refine_us", j["refine_us_per_iteration"], [(k["kernel"][9:36],k["us"]) for k in j["kernels"][:2]])
j=json.load(open("$O/uniform_$v.json")); print("$v uniform", j["wall_us_per_iteration"], [(k["kernel"][9:36],k["us"]) for k in j["kernel_table_us_per_iteration"][:2]])
PY
done
