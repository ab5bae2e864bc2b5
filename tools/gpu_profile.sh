#!/bin/bash
# On the GPU box: full rocprofv3 evidence of the default bench (S2, window-batched) + timelines of one step in both layouts.
# usage: tools/gpu_profile.sh TAG
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
tools/profile_round.sh $TAG
tools/pmc_passes.sh gpurun_out/prof_$TAG/sq
mkdir -p /tmp/prof_ref && cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_ref/stats -o bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-multi-stream --workload S2-ref-layout > $R/gpurun_out/prof_$TAG/bench_ref_under_rocprof.json 2> /dev/null
cd $R
python3 tools/timeline.py /tmp/prof_ref/stats gpurun_out/prof_$TAG/bench_ref_under_rocprof.json > gpurun_out/prof_$TAG/ref_layout_timeline.txt 2>&1
python3 tools/summarize_prof.py /tmp/prof_ref gpurun_out/prof_$TAG/ref_summary > /dev/null
tail -4 gpurun_out/prof_$TAG/timeline.txt; tail -3 gpurun_out/prof_$TAG/ref_layout_timeline.txt
head -14 gpurun_out/prof_$TAG/summary_kernel_stats.txt
