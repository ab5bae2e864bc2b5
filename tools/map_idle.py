#!/usr/bin/env python3
"""Live GPU idle and kernel table of one `SplatLoc.map` optimisation step (splatloc_amd.training.map_step; VERDICT r5 #4:
"live idle <= 50 us per step").  Same method as tools/refine_idle.py: wall time per step on an un-instrumented loop, GPU busy
time per step = the kernel durations of the same loop recorded by torch.profiler, idle = wall - busy; plus the kernel table
split into RASTER kernels (the C ABI's forward / backward stages) and everything else (activations, losses, statistics,
regulariser, Adam, densify, torch operators), so that "non-raster time of a map step" is a measured number.
usage: python tools/map_idle.py [workload=S2-ref-layout] [steps=100] [densify_every=0]"""
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

RASTER = ("preprocess_kernel", "preprocess_bwd_kernel", "composite_", "bin_walk", "bin_sort", "scan_", "sort_", "emit_kernel",
          "payload_kernel", "pad_features", "tile_order", "gather_dcolors", "ranges_", "radix", "onesweep", "depth_keys",
          "Memset", "fill_", "clear_")


def build(workload, dev):
    from splatloc_amd.camera import PinholeCamera
    from splatloc_amd.optim import Adam as FusedAdam
    from splatloc_amd.synthetic import WORKLOADS, make_workload
    wl = WORKLOADS[workload]
    sc = make_workload(workload)
    P0, W, H, C = wl["P"], wl["W"], wl["H"], wl["C"]
    E = max(C - 3, 1)
    g = torch.Generator().manual_seed(11)
    par = lambda t: torch.nn.Parameter(t.to(dev).contiguous().requires_grad_(True))  # noqa: E731
    names = ("xyz", "f_dc", "f_rest", "opacity", "marker", "kp_score", "scaling", "rotation")
    attr = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "marker": "_marker",
            "kp_score": "_kp_score", "scaling": "_scaling", "rotation": "_rotation"}
    # learning rates 0: the optimiser runs (same kernels, same traffic) but the scene does not move, so that every timed region
    # and the profiled loop see the SAME frame (with the reference's rates the lists of the synthetic scene shrink step by step:
    # the regions of one run went 7.9 -> 5.7 ms on S2, and "idle = wall - busy" compared different scenes)
    lr = {k: 0.0 for k in ("xyz", "f_dc", "f_rest", "opacity", "marker", "kp_score", "scaling", "rotation")}
    pc = types.SimpleNamespace(
        _xyz=par(sc.means3D.clone()), _features_dc=par(((sc.features[:, :3] - 0.5) / 0.28209479177387814)[:, None, :].contiguous()),
        _features_rest=par(torch.zeros(P0, 0, 3)), _opacity=par(torch.logit(sc.opacities.clamp(1e-4, 1 - 1e-4))),
        _marker=par((torch.rand(P0, 1, generator=g) < 0.05).float() * torch.rand(P0, 1, generator=g) * 0.9),
        _kp_score=par(torch.rand(P0, E, generator=g)), _scaling=par(torch.log(sc.scales)), _rotation=par(sc.rotations.clone()),
        active_sh_degree=0, max_sh_degree=0, percent_dense=0.01, primitive_reg=True, lr_init=0.0, lr_final=0.0,
        lr_delay_mult=0.01, max_steps=30000)
    pc.optimizer = FusedAdam([{"params": [getattr(pc, attr[k])], "lr": lr[k], "name": k} for k in names], lr=0.0, eps=1e-15)
    pc.xyz_gradient_accum = torch.zeros(P0, 1, device=dev)
    pc.denom = torch.zeros(P0, 1, device=dev)
    pc.max_radii2D = torch.zeros(P0, device=dev)
    g = torch.Generator().manual_seed(12)
    views = []
    for k in range(5):
        ang = torch.tensor(0.02 * (k - 2))
        R = torch.tensor([[torch.cos(ang), 0, torch.sin(ang)], [0, 1, 0], [-torch.sin(ang), 0, torch.cos(ang)]])
        cam = PinholeCamera(W, H, W / 2.0, W / 2.0, (W - 1) / 2.0, (H - 1) / 2.0, R, torch.tensor([0.01 * k, 0.0, 0.0])).to(dev)
        cam.original_image = torch.rand(3, H, W, generator=g).to(dev)
        cam.depth = (0.5 + 3 * torch.rand(H, W, generator=g)).to(dev)
        cam.kp_score = (torch.rand(H, W, generator=g) ** 4).to(dev)
        cam.exposure_a = torch.zeros(1, device=dev, requires_grad=True)
        cam.exposure_b = torch.zeros(1, device=dev, requires_grad=True)
        views.append(cam)
    return pc, views


def main():
    from splatloc_amd.training import map_step
    workload = sys.argv[1] if len(sys.argv) > 1 else "S2-ref-layout"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    every = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    dev = torch.device("cuda:0")
    pc, views = build(workload, dev)
    bg = torch.zeros(3, device=dev)
    pipe = types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False)
    cfg = {"Training": {"rgb_boundary_threshold": 0.01, "primitive_reg": True}}
    dens = dict(grad_threshold=0.0002, min_opacity=0.005, extent=6.0, size_threshold=20, every=every, offset=every // 3) if every else None
    it = [0]

    def loop(n):
        for _ in range(n):
            map_step(views, pc, pipe, bg, cfg, it[0], densify=dens, seed=7)
            it[0] += 1
            for cam in views:
                cam.exposure_a.grad = cam.exposure_b.grad = None

    saved = dens
    dens = None
    loop(10)
    dens = saved
    torch.cuda.synchronize(dev)
    rows0 = int(pc._xyz.shape[0])
    walls = []
    for _ in range(3 if every else 5):
        t0 = time.perf_counter()
        loop(N)
        torch.cuda.synchronize(dev)
        walls.append((time.perf_counter() - t0) / N * 1e6)
    rows1 = int(pc._xyz.shape[0])
    walls_sorted = sorted(walls)
    wall_us = walls_sorted[len(walls) // 2]
    t0 = time.perf_counter()
    loop(N)
    host_us = (time.perf_counter() - t0) / N * 1e6
    torch.cuda.synchronize(dev)
    from torch.profiler import ProfilerActivity, profile
    M = min(N, 40)
    t0 = time.perf_counter()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        loop(M)
        torch.cuda.synchronize(dev)
    rows = []
    for e in prof.key_averages():
        dt = getattr(e, "device_time_total", None)
        if dt is None:
            dt = getattr(e, "cuda_time_total", 0)
        if dt and e.count:
            rows.append((e.key[:90], e.count / M, dt / M))
    rows.sort(key=lambda r: -r[2])
    busy_us = sum(r[2] for r in rows)
    is_raster = lambda k: any(s in k for s in RASTER) and "adam" not in k and "activate" not in k  # noqa: E731
    raster_us = sum(u for k, c, u in rows if is_raster(k))
    # the profiled loop ran AFTER the timed regions: with densification the model is larger there, so idle is only quoted for a
    # constant-size run (every = 0)
    out = {"workload": workload, "steps_per_region": N, "densify_every": every, "rows_before_after_timed_regions": [rows0, rows1],
           "wall_us_per_step": round(wall_us, 1), "wall_us_all_regions": [round(w, 1) for w in walls],
           "host_enqueue_us_per_step": round(host_us, 1), "gpu_busy_us_per_step_torch_profiler": round(busy_us, 1),
           "idle_us_per_step": round(wall_us - busy_us, 1) if not every else None,
           "raster_kernels_us_per_step": round(raster_us, 1), "non_raster_kernels_us_per_step": round(busy_us - raster_us, 1),
           "kernels_per_step": round(sum(r[1] for r in rows), 1),
           "kernel_table_us_per_step": [{"kernel": k, "launches": round(c, 2), "us": round(u, 1), "raster": is_raster(k)} for k, c, u in rows[:60]]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
