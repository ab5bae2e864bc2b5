#!/usr/bin/env python3
"""Copies the evidence of a tools/gpu_round.sh run (gpurun_out/TAG/, gpurun_out/prof_TAG/) into the tracked profiles/ directory
as rNN_* files, then regenerates profiles/README.md.    usage: collect_round.py TAG rNN"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "collect_profiles.py"), tag, rnd], check=True, stdout=subprocess.DEVNULL)
prof = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
for a, b in (("ref_summary_kernel_stats.txt", "ref_layout_kernel_stats.txt"), ("ref_layout_timeline.txt", "ref_layout_timeline.txt"),
             ("bench_ref_under_rocprof.json", "bench_ref_layout_under_rocprof.json")):
    if os.path.exists(os.path.join(prof, a)):
        shutil.copy(os.path.join(prof, a), os.path.join(dst, f"{rnd}_{b}"))
names = {"bench_S2.json": "bench.json", "bench_S0.json": "bench_S0.json", "bench_S1.json": "bench_S1.json",
         "bench_S2-ref-layout.json": "bench_S2-ref-layout.json", "clocks.json": "clocks.json", "refine_step_timeline.txt": "refine_step_timeline.txt",
         "stage_eval_rendering.json": "stage_eval_rendering.json", "stage_map_step.json": "stage_map_step.json",
         "stage_map_step_ref.json": "stage_map_step_ref.json", "stage_refine_step_ref.json": "stage_refine_step_ref.json",
         "stage_pose_refine.json": "stage_pose_refine.json", "scene.json": "scene.json", "scene_replica_scale.json": "scene_replica_scale.json",
         "scene_radix_front_end.json": "scene_radix_front_end.json", "refine_idle.json": "refine_idle.json", "refine_idle_S0.json": "refine_idle_S0.json",
         "refine_idle_radix_front_end.json": "refine_idle_radix_front_end.json", "scene_lists.json": "scene_lists.json",
         "scene_lists_radix_front_end.json": "scene_lists_radix_front_end.json",
         "scene_lists_one_wave_forward.json": "scene_lists_one_wave_forward.json", "lone_wave.json": "lone_wave.json",
         "rccl_contact.json": "rccl_contact.json", "bench_force_process_group.json": "bench_force_process_group.json",
         "bench_force_process_group_rs_ag.json": "bench_force_process_group_rs_ag.json", "map_idle_S2-ref-layout.json": "map_idle_S2-ref-layout.json",
         "map_idle_S2.json": "map_idle_S2.json", "scene_lists_replica_scale.json": "scene_lists_replica_scale.json"}
for a, b in names.items():
    p = os.path.join(src, a)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, f"{rnd}_{b}"))
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "profiles_readme.py")], check=True)
print(sorted(f for f in os.listdir(dst) if f.startswith(rnd)))
