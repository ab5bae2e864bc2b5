#!/bin/bash
# round-4 final evidence (after the back-to-front backward): everything profiles/r04_* is built from
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r4z
O=gpurun_out/r4z
bash tools/gpu_profile.sh r04 > $O/profile.log 2>&1; tail -12 $O/profile.log
python tools/clock_trace.py $O/clocks.json > $O/clock_summary.txt 2>&1; cut -c1-400 $O/clock_summary.txt
python bench.py > $O/bench_S2.json 2> $O/bench_S2.err; cut -c1-200 $O/bench_S2.json; echo
for wl in S2-ref-layout S1 S0; do python bench.py --no-cpu-baseline --workload $wl > $O/bench_$wl.json 2>/dev/null; cut -c1-160 $O/bench_$wl.json; echo; done
python bench.py --stage eval_rendering --steps 5 --warmup 1 > $O/stage_eval.json 2>/dev/null
python bench.py --stage map_step > $O/stage_map_S2.json 2>/dev/null
python bench.py --stage map_step --workload S2-ref-layout --steps 100 --warmup 10 > $O/stage_map_ref.json 2>/dev/null
python bench.py --stage refine_step --workload S2-ref-layout --steps 300 --warmup 30 > $O/stage_refine_ref.json 2>/dev/null
python bench.py --stage scene > $O/scene.json 2>/dev/null; cut -c1-200 $O/scene.json; echo
python tools/refine_idle.py S2-ref-layout 300 > $O/refine_idle.json 2>/dev/null
python tools/refine_idle.py S0 300 > $O/refine_idle_S0.json 2>/dev/null
bash tools/ms_timeline.sh S2-ref-layout refine_step > /dev/null 2>&1; cp gpurun_out/map_step_timeline.txt $O/refine_step_timeline.txt
python tools/grad_bar_probe.py > $O/grad_bars.txt 2>&1; cp gpurun_out/r4_grad_bars.json $O/
ls $O
