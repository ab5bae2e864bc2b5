"""Per-phase cycle counters of the wide backward (SR_BWD_PROFILE, kept in tools/patches/r05_variants.patch):
    python tools/ablate.py --patch tools/patches/r05_variants.patch bprof "-DSR_BWD_PROFILE"   (here, before gpurun)
    python tools/bwd_prof.py [variant name = bprof] [workload = S2]                              (on the GPU box)

A stamp is an s_memtime plus the wait for it: ~40 cycles, a dozen of them per pair of Gaussians.  With every stamp on, the
stamps are themselves the largest consumer and whatever the phases do not cover looks like "not attributed" time (round 4:
16 % of a wave's life).  This tool therefore measures every phase ALONE (splatraster_debug_bwd_prof_select: only that phase's
two stamps execute), the whole life of a wave with NO phase stamp, and the all-stamps run for comparison; the kernel's
duration (torch.profiler device time; `bwd_prof.py base` gives the shipped library's) is printed for every selection so that the cost of the
instrumentation is visible."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, ".")
VARIANT = sys.argv[1] if len(sys.argv) > 1 else "bprof"     # "base": only the duration of the shipped library's launch sequence
os.environ["SPLATRASTER_LIB"] = os.path.abspath("splatloc_amd/_lib/libsplatraster.so" if VARIANT == "base" else
                                                "splatloc_amd/_lib/variants/libsplatraster_%s.so" % VARIANT)
from splatloc_amd import _native  # noqa: E402
from splatloc_amd.synthetic import make_workload  # noqa: E402
from tests.helpers import HipRun  # noqa: E402

lib = _native.load()


def bwd_kernel_us():
    """device time of the composite_bwd launches of one HipRun (torch.profiler: kernel durations, not host time)"""
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        HipRun(sc, backward=True)
        torch.cuda.synchronize()
    return round(sum(e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total
                     for e in prof.key_averages() if "composite_bwd_kernel" in e.key), 1)


workload = sys.argv[2] if len(sys.argv) > 2 else "S2"     # a named workload, or P,W,H,C[,noaux] (e.g. 500000,640,480,3,noaux: the refinement frame)
if "," in workload:
    from splatloc_amd.synthetic import make_scene
    f = workload.split(",")
    sc = make_scene(int(f[0]), int(f[1]), int(f[2]), int(f[3]), seed=2, scale_median=float(os.environ.get("SCALE_MEDIAN", "0.00627")))
    AUX = not (len(f) > 4 and f[4] == "noaux")
else:
    sc = make_workload(workload)
    AUX = True
_HipRun = HipRun
HipRun = lambda sc_, backward=True: _HipRun(sc_, backward=backward, use_depth=AUX, use_alpha=AUX)  # noqa: E731
raw = C.CDLL(os.environ["SPLATRASTER_LIB"])
out = (C.c_ulonglong * 17)()
if VARIANT == "base":
    HipRun(sc, backward=True)
    print(json.dumps({"workload": workload, "shipped_library_composite_bwd_us": [bwd_kernel_us() for _ in range(4)]}))
    raise SystemExit(0)
NAMES = {18: "set-up of the walk (slot tables, first chunk word requested) -> first chunk", 0: "chunk word wait + ballot",
         15: "chunk head after the ballot (prefetch issue) -> first round", 1: "staging round (records + feature rows -> LDS)",
         13: "pop of the next 2 / 4 candidates (slot loop head)", 7: "4x4x1 dot products", 2: "alpha evaluation of a pair",
         3: "T / A chain + weights", 4: "butterfly | reduce_e (moments on the matrix pipe)", 12: "atomic issue of the butterfly's value",
         5: "weight park", 6: "panel flush (MFMA + atomics)", 14: "end of a pair group -> next stamped point (loop back-edges, flush tests)",
         17: "drain: s_waitcnt vmcnt(0) for the wave's last atomics"}
ALL = sum(1 << i for i in NAMES)


def run(sel):
    raw.splatraster_debug_bwd_prof_select(C.c_uint(sel))
    HipRun(sc, backward=True)            # warm-up with this selection
    torch.cuda.synchronize()
    raw.splatraster_debug_bwd_prof(out, 1)
    us = bwd_kernel_us()
    raw.splatraster_debug_bwd_prof(out, 1)
    return list(out), us


v_none, us_none = run(0)
waves = max(1, v_none[16])             # the probe samples every 61st workgroup and counts the waves it recorded
w = round(v_none[11] / waves)
v_all, us_all = run(ALL)
res = {"workload": workload, "sampled_waves": waves, "whole_wave_cycles_no_phase_stamps": w,
       "whole_wave_cycles_all_stamps": round(v_all[11] / max(1, v_all[16])), "covered_cycles_all_stamps": round(v_all[0] / max(1, v_all[16])),
       "composite_bwd_us": {"no_phase_stamps": us_none, "all_stamps": us_all},
       "per_wave_counts": {"chunks": round(v_none[8] / waves, 1), "staging_rounds": round(v_none[9] / waves, 1), "pairs": round(v_none[10] / waves, 1)},
       "phases": {}}
total_alone = 0
for i, name in NAMES.items():
    v, us = run(1 << i)
    alone = v[0] / max(1, v[16])
    total_alone += alone
    res["phases"][name] = {"alone": round(alone), "whole_wave_with_this_phase_stamped": round(v[11] / max(1, v[16])), "composite_bwd_us": us}
res["sum_of_phases_measured_alone"] = round(total_alone)
res["not_attributed"] = w - res["sum_of_phases_measured_alone"]
res["not_attributed_frac"] = round(res["not_attributed"] / w, 4)
raw.splatraster_debug_bwd_prof_select(C.c_uint(ALL))
print(json.dumps(res, indent=1))
print(f"\nwhole life of a wave: {w} cycles without phase stamps, {res['whole_wave_cycles_all_stamps']} with all of them, of which "
      f"{res['covered_cycles_all_stamps']} inside a stamped phase (composite_bwd kernel {us_none:.1f} / {us_all:.1f} us)")
for name, d in res["phases"].items():
    print(f"  {name:84s} {d['alone']:8d}  {100.0 * d['alone'] / w:5.1f} %")
print(f"  {'not attributed (whole life - sum of the phases measured alone)':84s} {res['not_attributed']:8d}  {100.0 * res['not_attributed_frac']:5.1f} %")
