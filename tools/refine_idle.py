#!/usr/bin/env python3
"""Live GPU idle of one `color_refinement` iteration (VERDICT r3 task 6a: "quantify the live idle ... not under rocprof").

Wall time per iteration is measured on an un-instrumented loop (N iterations between two synchronisations).  GPU busy time
per iteration is the sum of the kernel durations of the SAME loop recorded by torch.profiler (roctracer GPU time stamps of
every kernel in the process, also the ones launched through the C ABI).  idle = wall - busy.  Also prints the kernel table
of one iteration (name, count, total us) so that the figure can be held against profiles/rNN_refine_step_timeline.txt.
usage: python tools/refine_idle.py [workload=S2-ref-layout] [iterations=300]"""
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    from splatloc_amd.camera import PinholeCamera
    from splatloc_amd.optim import Adam as FusedAdam
    from splatloc_amd.synthetic import WORKLOADS, make_workload
    from splatloc_amd.training import color_refinement_step
    workload = sys.argv[1] if len(sys.argv) > 1 else "S2-ref-layout"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    dev = torch.device("cuda:0")
    wl = WORKLOADS[workload]
    sc = make_workload(workload)
    P0, W, H, C = wl["P"], wl["W"], wl["H"], wl["C"]
    E = max(C - 3, 1)
    g = torch.Generator().manual_seed(11)
    par = lambda t: torch.nn.Parameter(t.to(dev).contiguous().requires_grad_(True))  # noqa: E731
    names = ("xyz", "f_dc", "f_rest", "opacity", "marker", "kp_score", "scaling", "rotation")
    attr = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "marker": "_marker",
            "kp_score": "_kp_score", "scaling": "_scaling", "rotation": "_rotation"}
    lr = {"xyz": 1.6e-4 * 6.0, "f_dc": 2.5e-3, "f_rest": 2.5e-3 / 20, "opacity": 5e-2, "marker": 5e-2, "kp_score": 5e-2,
          "scaling": 1e-3 * 6.0, "rotation": 1e-3}
    pc = types.SimpleNamespace(
        _xyz=par(sc.means3D.clone()), _features_dc=par(((sc.features[:, :3] - 0.5) / 0.28209479177387814)[:, None, :].contiguous()),
        _features_rest=par(torch.zeros(P0, 0, 3)), _opacity=par(torch.logit(sc.opacities.clamp(1e-4, 1 - 1e-4))),
        _marker=par((torch.rand(P0, 1, generator=g) < 0.05).float() * torch.rand(P0, 1, generator=g) * 0.9),
        _kp_score=par(torch.rand(P0, E, generator=g)), _scaling=par(torch.log(sc.scales)), _rotation=par(sc.rotations.clone()),
        active_sh_degree=0, max_sh_degree=0, lr_init=1.6e-4 * 6.0, lr_final=1.6e-6 * 6.0, lr_delay_mult=0.01, max_steps=30000)
    pc.optimizer = FusedAdam([{"params": [getattr(pc, attr[k])], "lr": lr[k], "name": k} for k in names], lr=0.0, eps=1e-15)
    pc.max_radii2D = torch.zeros(P0, device=dev)
    views = []
    for k in range(8):
        ang = torch.tensor(0.02 * (k - 4))
        R = torch.tensor([[torch.cos(ang), 0, torch.sin(ang)], [0, 1, 0], [-torch.sin(ang), 0, torch.cos(ang)]])
        cam = PinholeCamera(W, H, W / 2.0, W / 2.0, (W - 1) / 2.0, (H - 1) / 2.0, R, torch.tensor([0.01 * k, 0.0, 0.0])).to(dev)
        cam.original_image = torch.rand(3, H, W, generator=g).to(dev)
        views.append(cam)
    bg = torch.zeros(3, device=dev)
    pipe = types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False)
    it = [0]

    def loop(n):
        for _ in range(n):
            it[0] += 1
            color_refinement_step(views[it[0] % 8], pc, pipe, bg, 0.2, it[0])

    loop(50)
    torch.cuda.synchronize(dev)
    walls = []
    for _ in range(5):
        t0 = time.perf_counter()
        loop(N)
        torch.cuda.synchronize(dev)
        walls.append((time.perf_counter() - t0) / N * 1e6)
    walls.sort()
    wall_us = walls[len(walls) // 2]
    # host-only cost: the same loop with every launch still enqueued but timed WITHOUT the final wait dominates only when the host is
    # the bottleneck; report the enqueue time of N iterations as well (perf_counter before the synchronize)
    t0 = time.perf_counter()
    loop(N)
    host_us = (time.perf_counter() - t0) / N * 1e6
    torch.cuda.synchronize(dev)
    from torch.profiler import ProfilerActivity, profile
    M = 60
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        loop(M)
        torch.cuda.synchronize(dev)
    rows = []
    for e in prof.key_averages():
        dt = getattr(e, "device_time_total", None)
        if dt is None:
            dt = getattr(e, "cuda_time_total", 0)
        if dt and e.count:
            rows.append((e.key[:70], e.count / M, dt / M))
    rows.sort(key=lambda r: -r[2])
    busy_us = sum(r[2] for r in rows)
    out = {"workload": workload, "iterations_per_region": N, "wall_us_per_iteration": round(wall_us, 1),
           "wall_us_all_regions": [round(w, 1) for w in walls], "host_enqueue_us_per_iteration": round(host_us, 1),
           "gpu_busy_us_per_iteration_torch_profiler": round(busy_us, 1), "idle_us_per_iteration": round(wall_us - busy_us, 1),
           "kernels_per_iteration": round(sum(r[1] for r in rows), 1),
           "kernel_table_us_per_iteration": [{"kernel": k, "launches": round(c, 2), "us": round(u, 1)} for k, c, u in rows[:40]]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
