#!/usr/bin/env python3
"""How fast ONE wave walks a long tile list when it runs (almost) alone on its SIMD — the regime that sets the duration of the
compositing kernels on a single 640x480 frame (4 800 quadrant waves start at once; the kernel lasts as long as the longest list).
A 16x16 image (one tile, four quadrant waves) with N translucent Gaussians in front of it: the list has ~N entries, no pixel
saturates, and the kernel's duration is the walk of that list.  Prints microseconds per 1 000 list entries (slope between the
two largest N) for the forward and the backward of a C-channel layout, with one wave per quadrant and with the teams of four
waves that narrow layouts get for their longest lists (composite_fwd.hip).
usage: python tools/lone_wave.py [C=4] [N ...=1000 2000 4000]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from torch.profiler import ProfilerActivity, profile

    from splatloc_amd import _native
    from splatloc_amd.synthetic import make_scene
    from tests.helpers import HipRun
    lib = _native.load()
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    Ns = [int(a) for a in sys.argv[2:]] or [1000, 2000, 4000]
    out = {"channels": C}
    for mode, label in ((0, "one_wave_per_quadrant"), (-1, "default")):
        lib.splatraster_debug_set_fwd_team(mode)
        out[label] = measure(C, Ns, profile, ProfilerActivity, make_scene, HipRun)
    lib.splatraster_debug_set_fwd_team(-1)
    print(json.dumps(out, indent=1))


def measure(C, Ns, profile, ProfilerActivity, make_scene, HipRun):
    rows = []
    for N in Ns:
        sc = make_scene(N, 16, 16, C, seed=5, scale_median=0.25)
        sc.opacities = sc.opacities * 0.05          # alpha <= 0.05: ~300 hits before a pixel's transmittance falls below 1e-4
        run = HipRun(sc, backward=True)
        R = int(run.num_rendered)
        best = {}
        for _ in range(5):
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                HipRun(sc, backward=True)
                torch.cuda.synchronize()
            for e in prof.key_averages():
                t = e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total
                for key in ("composite_fwd", "composite_bwd"):
                    if key in e.key:
                        best[key] = min(best.get(key, 1e30), t)
        n_contrib = run.state["n_contrib"] if "n_contrib" in run.state else None
        rows.append({"N": N, "list_entries": R, "fwd_us": round(best.get("composite_fwd", 0.0), 1), "bwd_us": round(best.get("composite_bwd", 0.0), 1),
                     "max_contributors": int(n_contrib.max()) if n_contrib is not None else None})
    out = {"runs": rows}
    if len(rows) >= 2:
        a, b = rows[-2], rows[-1]
        d = max(1, b["list_entries"] - a["list_entries"]) / 1000.0
        out["fwd_us_per_1000_entries"] = round((b["fwd_us"] - a["fwd_us"]) / d, 2)
        out["bwd_us_per_1000_entries"] = round((b["bwd_us"] - a["bwd_us"]) / d, 2)
    return out


if __name__ == "__main__":
    main()
