#!/bin/bash
# Decision-grade A/B of the color_refinement iteration on a reconstructed Replica-scale scene (180 key-frames, 413k rows), ONE box, interleaved:
# every process rebuilds the scene, then times SCENE_LISTS_REGIONS regions of 300 iterations (the median is compared) — the whole-schedule bench
# (bench.py --stage scene) has a 3 % run-to-run spread and cannot resolve a 4 % change.  Round 6 used it for: the shipped library (base) | the long-list
# sort launch on the side stream (forked) | 4 parts for every list (k0: tools/ablate.py k0 "-DSR_SPLIT_EXTRA_TILES=0") | both (k0_forked);
# profiles/r06_ab_probes.txt #6.        usage (on the GPU box): tools/ab_refine.sh [OUTDIR] [ROUNDS]
O=${1:-gpurun_out/ab_refine}; ROUNDS=${2:-4}; mkdir -p $O
V=$PWD/splatloc_amd/_lib/variants
export SCENE_LISTS_REGIONS=9
for r in $(seq 1 $ROUNDS); do
  for v in base forked k0 k0_forked; do
    unset SPLATRASTER_LIB SPLATRASTER_SORT_FORK
    case $v in k0*) [ -f $V/libsplatraster_k0.so ] || continue; export SPLATRASTER_LIB=$V/libsplatraster_k0.so;; esac
    case $v in *forked) export SPLATRASTER_SORT_FORK=-1;; esac
    python tools/scene_lists.py 180 600000 300 > $O/replica_${v}_$r.json 2>/dev/null
    python -c "
import json; j=json.load(open('$O/replica_${v}_$r.json')); print('$v $r', j['refine_us_per_iteration'], j['refine_us_all_regions'], [(k['kernel'][9:30],k['us']) for k in j['kernels'][:2]])"
  done
done
