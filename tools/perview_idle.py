#!/usr/bin/env python3
"""What the literal drop-in loop (one GaussianRasterizer call per view, train_gaussians.py:195-229) loses against the window path:
wall time per view of an un-instrumented loop of 5 per-view fwd+bwd calls vs the GPU busy time of the same loop (torch.profiler
kernel time stamps), i.e. how much of the gap is an idle GPU (the host read of R, Python between the calls) and how much is GPU
work the window path does not have (per-view depth sorts, autograd's gradient accumulation kernels).
usage: python tools/perview_idle.py [workload=S2-ref-layout] [steps=40]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from splatloc_amd import GaussianRasterizationSettings, GaussianRasterizer, rasterize_window
    from splatloc_amd.camera import PinholeCamera
    from splatloc_amd.synthetic import make_workload
    workload = sys.argv[1] if len(sys.argv) > 1 else "S2-ref-layout"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    dev = torch.device("cuda:0")
    sc = make_workload(workload).to(dev)
    cam0 = sc.camera
    W, H = cam0.image_width, cam0.image_height
    views = []
    for k in range(5):
        ang = 0.02 * (k - 2)
        R = torch.tensor([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], dtype=torch.float32)
        cam = PinholeCamera(W, H, cam0.fx, cam0.fy, cam0.cx, cam0.cy, R, torch.tensor([0.01 * k, 0.0, 0.0])).to(dev)
        rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, sc.bg, 1.0, cam.world_view_transform,
                                           cam.full_proj_transform, 0, cam.camera_center, False, False)
        views.append((GaussianRasterizer(raster_settings=rs), rs))
    leaf = lambda t: t.clone().requires_grad_(True)  # noqa: E731
    params = [leaf(sc.means3D), leaf(sc.features), leaf(sc.opacities), leaf(sc.scales), leaf(sc.rotations)]
    g = (sc.dL_dcolor, sc.dL_ddepth, sc.dL_dalpha)

    def per_view(n):
        for _ in range(n):
            for p in params:
                p.grad = None
            for rast, _ in views:
                m2 = torch.zeros_like(params[0], requires_grad=True)
                out = rast(means3D=params[0], means2D=m2, shs=None, colors_precomp=params[1], opacities=params[2],
                           scales=params[3], rotations=params[4], cov3D_precomp=None)
                torch.autograd.backward(out[:3], g)

    def window(n):
        for _ in range(n):
            for p in params:
                p.grad = None
            m2 = [torch.zeros_like(params[0], requires_grad=True) for _ in views]
            outs = rasterize_window([rs for _, rs in views], params[0], m2, params[1], params[2], scales=params[3], rotations=params[4])
            torch.autograd.backward([t for o in outs for t in o[:3]], [t for _ in views for t in g])

    from torch.profiler import ProfilerActivity, profile
    res = {"workload": workload, "views_per_step": 5}
    for name, fn in (("per_view_loop", per_view), ("window", window)):
        fn(5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(steps)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / steps * 1e6
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            fn(10)
            torch.cuda.synchronize()
        busy, torch_us, n_k = 0.0, 0.0, 0.0
        for e in prof.key_averages():
            dt = getattr(e, "device_time_total", None) or getattr(e, "cuda_time_total", 0)
            if dt and e.count:
                busy += dt / 10
                n_k += e.count / 10
                if "at::" in e.key or "Memcpy" in e.key or "Memset" in e.key:
                    torch_us += dt / 10
        res[name] = {"wall_us_per_step": round(wall, 1), "gpu_busy_us_per_step": round(busy, 1), "idle_us_per_step": round(wall - busy, 1),
                     "kernels_per_step": round(n_k, 1), "torch_and_copy_kernels_us_per_step": round(torch_us, 1),
                     "frames_per_s": round(5e6 / wall, 1)}
    a, b = res["per_view_loop"], res["window"]
    res["gap_us_per_step"] = round(a["wall_us_per_step"] - b["wall_us_per_step"], 1)
    res["of_which_idle_gpu_us"] = round(a["idle_us_per_step"] - b["idle_us_per_step"], 1)
    res["of_which_more_gpu_work_us"] = round(a["gpu_busy_us_per_step"] - b["gpu_busy_us_per_step"], 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
