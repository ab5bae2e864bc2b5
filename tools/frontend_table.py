#!/usr/bin/env python3
"""Kernel table (torch.profiler GPU time stamps) of the FORWARD front end of a window of V views.
usage: python tools/frontend_table.py [workload=S2-ref-layout] [V=1] [iterations=200]
The front end follows SPLATRASTER_FRONT_END (-1 default, 0 radix sorts, 1 binned)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from splatloc_amd import GaussianRasterizationSettings, rasterize_window
    from splatloc_amd.camera import PinholeCamera
    from splatloc_amd.synthetic import make_workload
    workload = sys.argv[1] if len(sys.argv) > 1 else "S2-ref-layout"
    V = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    dev = torch.device("cuda:0")
    sc = make_workload(workload)
    cam0 = sc.camera
    W, H = cam0.image_width, cam0.image_height
    settings = []
    for k in range(V):
        ang = 0.02 * (k - V // 2)
        R = torch.tensor([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], dtype=torch.float32)
        cam = PinholeCamera(W, H, cam0.fx, cam0.fy, cam0.cx, cam0.cy, R, torch.tensor([0.01 * k, 0.0, 0.0])).to(dev)
        settings.append(GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, sc.bg.to(dev), 1.0, cam.world_view_transform,
                                                      cam.full_proj_transform, 0, cam.camera_center, False, False))
    m3, col, opa, sca, rot = (t.to(dev) for t in (sc.means3D, sc.features, sc.opacities, sc.scales, sc.rotations))
    m2 = [torch.zeros_like(m3) for _ in settings]

    def loop(n):
        with torch.no_grad():
            for _ in range(n):
                rasterize_window(settings, m3, m2, col, opa, sca, rot)

    loop(20)
    torch.cuda.synchronize(dev)
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        loop(N)
        torch.cuda.synchronize(dev)
    rows = []
    for e in prof.key_averages():
        dt = getattr(e, "device_time_total", None)
        if dt is None:
            dt = getattr(e, "cuda_time_total", 0)
        if dt and e.count:
            rows.append((e.key[:60], e.count / N, dt / N))
    rows.sort(key=lambda r: -r[2])
    print(json.dumps({"workload": workload, "views": V, "front_end": os.environ.get("SPLATRASTER_FRONT_END", "-1"),
                      "lib": os.environ.get("SPLATRASTER_LIB", "in-tree"),
                      "busy_us": round(sum(r[2] for r in rows), 1),
                      "kernels": [{"kernel": k, "launches": round(c, 2), "us": round(u, 1)} for k, c, u in rows]}))
    for k, c, u in rows:
        print("   %-60s %5.2f %8.1f" % (k, c, u), file=sys.stderr)


if __name__ == "__main__":
    main()
