#!/bin/bash
# refinement iteration at Replica scale, interleaved A/B on one box: round-6 library | 4 parts for every list (k0) | sort launches serial | both reverted (= round 5)
O=gpurun_out/r06i; mkdir -p $O
V=$PWD/splatloc_amd/_lib/variants
export SCENE_LISTS_REGIONS=9
for r in 1 2 3 4; do
  for v in base k0 serial r05like; do
    unset SPLATRASTER_LIB SPLATRASTER_SORT_FORK
    [ $v = k0 ] && export SPLATRASTER_LIB=$V/libsplatraster_k0.so
    [ $v = serial ] && export SPLATRASTER_SORT_FORK=0
    [ $v = r05like ] && export SPLATRASTER_LIB=$V/libsplatraster_k0.so SPLATRASTER_SORT_FORK=0
    python tools/scene_lists.py 180 600000 300 > $O/replica_${v}_$r.json 2>/dev/null
    python -c "
import json; j=json.load(open('$O/replica_${v}_$r.json')); print('$v $r', j['refine_us_per_iteration'], j['refine_us_all_regions'], [(k['kernel'][9:30],k['us']) for k in j['kernels'][:2]])"
  done
done
SPLATRASTER_FRONT_END=1 python bench.py --no-cpu-baseline --no-multi-stream > $O/bench_S2_binned.json 2>/dev/null; python -c "
import json; j=json.load(open('$O/bench_S2_binned.json')); print('S2 binned', j['value'], {k:v['avg_ms'] for k,v in j['stages'].items()})"
python bench.py --no-cpu-baseline --no-multi-stream > $O/bench_S2_radix.json 2>/dev/null; python -c "
import json; j=json.load(open('$O/bench_S2_radix.json')); print('S2 radix', j['value'], {k:v['avg_ms'] for k,v in j['stages'].items()})"
