#!/bin/bash
for lib in "" f2bfix; do
  if [ -n "$lib" ]; then export SPLATRASTER_LIB=$(pwd)/splatloc_amd/_lib/variants/libsplatraster_$lib.so; fi
  echo "== lib: ${lib:-base}"
  python bench.py --stage map_step --workload S2-ref-layout --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('ref', j['ms_per_step'], j['densify']['ms_per_step_without_densify'], j['torch_front_end_loss_adam_same_rasterizer_no_densify'])"
  python bench.py --stage map_step --workload S2-ref-layout --steps 100 --warmup 10 --densify-every 0 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('ref no densify', j['ms_per_step'])"
done
