#!/bin/bash
# scene-level A/B on ONE box, interleaved: round-6 library vs the same with 4 parts for every list (k0) vs sort launches serial
O=gpurun_out/r06h; mkdir -p $O
V=$PWD/splatloc_amd/_lib/variants
timeout 600 python -m pytest tests/test_gpu_rccl.py -x -q 2>&1 | tail -2
for f in ring rs_ag; do python bench.py --gpus 1 --force-process-group --reduce $f --no-cpu-baseline --no-multi-stream > $O/bench_fpg_$f.json 2>/dev/null; python -c "
import json; j=json.load(open('$O/bench_fpg_$f.json')); print('force-process-group $f', j['value'], j['config']['grad_allreduce_path'])"; done
for r in 1 2 3; do
  for v in base k0 serial; do
    unset SPLATRASTER_LIB SPLATRASTER_SORT_FORK
    [ $v = k0 ] && export SPLATRASTER_LIB=$V/libsplatraster_k0.so
    [ $v = serial ] && export SPLATRASTER_SORT_FORK=0
    python bench.py --stage scene --keyframes 180 --truth 600000 > $O/scene_replica_${v}_$r.json 2>/dev/null
    python -c "
import json; j=json.load(open('$O/scene_replica_${v}_$r.json')); print('$v $r', j['ms_per_step'], j['map_ms_per_iteration'], j['refine_ms_per_iteration'])"
  done
done
