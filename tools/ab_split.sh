python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_refine.py tests/test_gpu_window.py tests/test_gpu_lineage_spec.py -x -q -k "not dist2 and not radix" 2>&1 | tail -4
for w in 0 8192 0 8192; do  # knob 0 = no split
  SPLATRASTER_SPLIT_MAX_WAVES=$w python bench.py --stage refine_step --workload S2-ref-layout --steps 60 --warmup 10 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('refine split<=$w', j['value'], j['ms_per_step'])"
done
for wl in S2-ref-layout S1-640 S0; do for w in 0 8192; do
  SPLATRASTER_SPLIT_MAX_WAVES=$w python bench.py --no-cpu-baseline --no-multi-stream --no-window --workload $wl | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('per-view loop $wl split<=$w', j['value'], {k: v['avg_ms'] for k, v in j['stages'].items() if 'composite' in k})"
done; done
