#!/usr/bin/env python3
"""Copies the summaries of a tools/profile_round.sh + tools/pmc_passes.sh run (gpurun_out/prof_TAG) into the
tracked profiles/ directory as rNN_* files and refreshes profiles/traffic.json (HBM bytes per launch of every
kernel group, read by bench.py as roofline.traffic) and profiles/valu.json (VALU / MFMA / issue statistics of the
two compositing kernels, read by bench.py as frame_valu / roofline_valu).  Both files record the workload and the launch
mode (frames per launch) they were measured on: bench.py attaches them only to a run of the same kind.
usage: collect_profiles.py TAG rNN [WORKLOAD=S2] [FRAMES_PER_LAUNCH=5]"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "S2"
fpl = int(sys.argv[4]) if len(sys.argv) > 4 else 5
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
for a, b in (("summary_kernel_stats.txt", "kernel_stats.txt"), ("summary_pmc.json", "pmc_hbm.json"), ("timeline.txt", "timeline.txt"),
             ("bench_under_rocprof.json", "bench_under_rocprof.json"), ("sq/pmc_summary.json", "pmc_sq_counters.json")):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, f"{rnd}_{b}"))
hbm = json.load(open(os.path.join(dst, f"{rnd}_pmc_hbm.json")))
group = {"composite_bwd": ["composite_bwd_kernel"], "composite_fwd": ["composite_fwd_kernel"], "emit": ["emit_kernel"],
         "preprocess": ["preprocess_kernel"], "preprocess_bwd": ["preprocess_bwd_kernel", "gather_dcolors_kernel"],
         "payload": ["payload_kernel", "pad_features_kernel"], "tile_sort": ["sort_hist_kernel", "sort_scatter_kernel", "scan_onepass_kernel<true>"],
         "depth_sort": ["sort_hist_all_kernel", "sort_sweep_kernel"], "scan": ["scan_onepass_kernel<false>"]}
traffic = {}
for g, keys in group.items():
    tot = 0
    for k, e in hbm.items():
        if any(k.startswith(p) for p in keys) and "hbm_bytes_per_launch" in e:
            if g == "preprocess" and k.startswith("preprocess_bwd"):
                continue
            # launches per frame: kernels launched several times per frame are summed per frame
            per_frame = e["FETCH_SIZE_launches"] / max(hbm["composite_fwd_kernel<35>"]["FETCH_SIZE_launches"], 1) \
                if "composite_fwd_kernel<35>" in hbm else 1
            tot += int(e["hbm_bytes_per_launch"] * per_frame)
    if tot:
        traffic[g] = tot
traffic["_workload"] = workload
traffic["_frames_per_launch"] = fpl
traffic["_note"] = (f"HBM bytes per launch sequence ({fpl} frame(s): one window) of each stage from rocprofv3 PMC passes on {workload} "
                    f"(profiles/{rnd}_pmc_hbm.json): 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE, KiB -> bytes; one launch per "
                    "sequence for the compositing kernels")
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
sq = json.load(open(os.path.join(dst, f"{rnd}_pmc_sq_counters.json")))
valu = {"_workload": workload, "_frames_per_launch": fpl,
        "_note": f"rocprofv3 SQ counters per launch ({fpl} frame(s)) on {workload} (profiles/{rnd}_pmc_sq_counters.json, summed over the 8 XCDs); "
                 "valu_busy = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8; "
                 "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * kernel cycles)"}
for k, c in sq.items():
    if "composite" not in k or "GRBM_GUI_ACTIVE" not in c:
        continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    valu[k] = {"kernel_cycles": int(cyc), "valu_wave_instructions": int(c.get("SQ_INSTS_VALU", 0) - c.get("SQ_INSTS_MFMA", 0)),
               "mfma_wave_instructions": int(c.get("SQ_INSTS_MFMA", 0)), "salu_wave_instructions": int(c.get("SQ_INSTS_SALU", 0)),
               "lds_wave_instructions": int(c.get("SQ_INSTS_LDS", 0)),
               "valu_busy": round(c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (1024 * cyc), 4),
               "mfma_busy": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * cyc), 4),
               "wave_cycles_waiting_any": round(c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1), 4),
               "wave_cycles_waiting_issue": round(c.get("SQ_WAIT_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1), 4),
               "lds_bank_conflict_cycles": int(c.get("SQ_LDS_BANK_CONFLICT", 0)), "waves": int(c.get("SQ_WAVES", 0))}
json.dump(valu, open(os.path.join(dst, "valu.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
print(json.dumps(valu, indent=1))
