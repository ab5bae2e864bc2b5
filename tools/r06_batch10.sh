#!/bin/bash
O=gpurun_out/r06j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_binsort.py -x -q 2>&1 | tail -2
for cap in 0 1024; do
  for r in 1 2 3; do
  SPLATRASTER_FRONT_END=1 SPLATRASTER_TILE_SORT_CAP=$cap python bench.py --no-cpu-baseline --no-multi-stream > $O/bench_S2_binned_cap$cap.json 2>/dev/null; python -c "
import json; j=json.load(open('$O/bench_S2_binned_cap$cap.json')); print('S2 binned cap $cap', j['value'], {k:v['avg_ms'] for k,v in j['stages'].items() if k in ('depth_sort','tile_sort','payload')})"
  done
done
for r in 1 2 3; do python bench.py --no-cpu-baseline --no-multi-stream > $O/bench_S2_radix.json 2>/dev/null; python -c "
import json; j=json.load(open('$O/bench_S2_radix.json')); print('S2 radix', j['value'])"; done
for cap in 0 1024; do
  SPLATRASTER_FRONT_END=1 SPLATRASTER_TILE_SORT_CAP=$cap python bench.py --no-window --no-cpu-baseline --no-multi-stream > $O/bench_S2_perview_binned_cap$cap.json 2>/dev/null; python -c "
import json; j=json.load(open('$O/bench_S2_perview_binned_cap$cap.json')); print('S2 per-view binned cap $cap', j['value'], {k:v['avg_ms'] for k,v in j['stages'].items() if k in ('depth_sort','tile_sort','payload')})"
done
for cap in 0 1024; do SPLATRASTER_TILE_SORT_CAP=$cap python bench.py --no-cpu-baseline --no-multi-stream --workload S1 > $O/bench_S1_cap$cap.json 2>/dev/null; python -c "
import json; j=json.load(open('$O/bench_S1_cap$cap.json')); print('S1 cap $cap', j['value'], {k:v['avg_ms'] for k,v in j['stages'].items() if k in ('depth_sort','tile_sort','payload')})"; done
