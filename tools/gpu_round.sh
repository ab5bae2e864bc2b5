#!/bin/bash
# The round's evidence in one run on the GPU box (replaces the per-round scratch scripts):
#   tools/gpu_round.sh TAG [all|profile|bench|stages|scene|dist|idle]      e.g.  tools/gpu_round.sh r06 all
# Everything is written under gpurun_out/TAG/; tools/collect_round.py TAG rNN copies the summaries into profiles/.
TAG=${1:-r05}; WHAT=${2:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/$TAG
mkdir -p $O
want() { [ "$WHAT" = all ] || [ "$WHAT" = "$1" ]; }
if want profile; then
  bash tools/gpu_profile.sh $TAG > $O/profile.log 2>&1; tail -6 $O/profile.log
  python tools/clock_trace.py $O/clocks.json > $O/clock_summary.txt 2>&1; cut -c1-300 $O/clock_summary.txt
  # (no rocprofv3 timeline of the refinement iteration any more: the profiler's own overhead made it 524 us of a 398-us iteration;
  #  the LIVE figures are tools/refine_idle.py's, below)
fi
if want bench; then
  python bench.py > $O/bench_S2.json 2> $O/bench_S2.err; cut -c1-200 $O/bench_S2.json; echo
  python bench.py --workload S0 > $O/bench_S0.json 2>/dev/null; cut -c1-160 $O/bench_S0.json; echo
  for wl in S2-ref-layout S1; do python bench.py --no-cpu-baseline --workload $wl > $O/bench_$wl.json 2>/dev/null; cut -c1-160 $O/bench_$wl.json; echo; done
fi
if want stages; then
  python bench.py --stage eval_rendering --steps 5 --warmup 1 > $O/stage_eval_rendering.json 2>/dev/null
  python bench.py --stage map_step > $O/stage_map_step.json 2>/dev/null
  python bench.py --stage map_step --workload S2-ref-layout --steps 100 --warmup 10 > $O/stage_map_step_ref.json 2>/dev/null
  python bench.py --stage refine_step --workload S2-ref-layout --steps 300 --warmup 30 > $O/stage_refine_step_ref.json 2>/dev/null
  python bench.py --stage pose_refine > $O/stage_pose_refine.json 2>/dev/null
  for f in $O/stage_*.json; do cut -c1-200 $f; echo; done
fi
if want scene; then
  python bench.py --stage scene > $O/scene.json 2>/dev/null; cut -c1-200 $O/scene.json; echo
  python bench.py --stage scene --keyframes 180 --truth 600000 > $O/scene_replica_scale.json 2>/dev/null; cut -c1-200 $O/scene_replica_scale.json; echo
  SPLATRASTER_FRONT_END=0 python bench.py --stage scene > $O/scene_radix_front_end.json 2>/dev/null; cut -c1-200 $O/scene_radix_front_end.json; echo
fi
if want dist; then
  python tools/rccl_contact.py > $O/rccl_contact.json 2>$O/rccl_contact.err; cut -c1-200 $O/rccl_contact.json; echo
  python bench.py --gpus 1 --force-process-group --no-cpu-baseline --no-multi-stream > $O/bench_force_process_group.json 2>/dev/null; cut -c1-160 $O/bench_force_process_group.json; echo
  python bench.py --gpus 1 --force-process-group --reduce rs_ag --no-cpu-baseline --no-multi-stream > $O/bench_force_process_group_rs_ag.json 2>/dev/null; cut -c1-160 $O/bench_force_process_group_rs_ag.json; echo
fi
if want idle; then
  python tools/map_idle.py S2-ref-layout 100 0 > $O/map_idle_S2-ref-layout.json 2>/dev/null; cut -c1-300 $O/map_idle_S2-ref-layout.json; echo
  python tools/map_idle.py S2 40 0 > $O/map_idle_S2.json 2>/dev/null; cut -c1-300 $O/map_idle_S2.json; echo
  python tools/scene_lists.py 180 600000 300 > $O/scene_lists_replica_scale.json 2>/dev/null; cut -c1-200 $O/scene_lists_replica_scale.json; echo
  python tools/refine_idle.py S2-ref-layout 300 > $O/refine_idle.json 2>/dev/null; cut -c1-200 $O/refine_idle.json; echo
  python tools/refine_idle.py S0 300 > $O/refine_idle_S0.json 2>/dev/null
  SPLATRASTER_FRONT_END=0 python tools/refine_idle.py S2-ref-layout 300 > $O/refine_idle_radix_front_end.json 2>/dev/null
  python tools/scene_lists.py 60 200000 300 > $O/scene_lists.json 2>/dev/null
  SPLATRASTER_FRONT_END=0 python tools/scene_lists.py 60 200000 300 > $O/scene_lists_radix_front_end.json 2>/dev/null
  SPLATRASTER_FWD_TEAM=0 python tools/scene_lists.py 60 200000 300 > $O/scene_lists_one_wave_forward.json 2>/dev/null
  python tools/lone_wave.py 4 1000 2000 4000 > $O/lone_wave.json 2>/dev/null
fi
ls $O
