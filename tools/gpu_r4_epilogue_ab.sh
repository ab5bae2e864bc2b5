#!/bin/bash
# A/B of two builds of the library over 10-second clock traces and the default bench (profiles/r04_ab_probes.txt item 12: the forward's
# epilogue with back-to-back stores vs the shipped one).  $V = the variant library built beforehand with tools/ablate.py (here: the
# shipped sources as "preep" while the working tree held the candidate); run on the GPU box: bash tools/gpu_r4_epilogue_ab.sh
V=splatloc_amd/_lib/variants/libsplatraster_preep.so
mkdir -p gpurun_out/epab
for i in 1 2; do
python tools/clock_trace.py gpurun_out/epab/new_$i.json > /dev/null 2>&1
SPLATRASTER_LIB=$V python tools/clock_trace.py gpurun_out/epab/old_$i.json > /dev/null 2>&1
done
for i in 1 2 3; do
python bench.py --no-cpu-baseline | cut -c1-180
SPLATRASTER_LIB=$V python bench.py --no-cpu-baseline | cut -c1-180
done
python - <<'PY'
import json,glob
for p in sorted(glob.glob('gpurun_out/epab/*.json')):
    j=json.load(open(p)); a=j['amdsmi_library_20hz']; r=a['residency_counters_first_last']
    print(p.split('/')[-1], j['bench_value_frames_per_s'], j['bench_repeats']['ms_per_step_all'], 'clk', a['gfxclk_mhz_mean_over_xcds']['median'], 'W', a['socket_power_w']['median'], 'ppt', round((r['ppt_residency_acc'][1]-r['ppt_residency_acc'][0])/max(r['accumulation_counter'][1]-r['accumulation_counter'][0],1),2))
PY
