#!/usr/bin/env python3
"""Where the HOST time of one optimisation step goes (the refinement iteration is host-bound: tools/refine_idle.py):
cProfile of N color_refinement_step / map_step calls at the reference layout, sorted by cumulative and by own time.
usage: python tools/hostprof_steps.py [refine|map] [workload=S0] [N=200]"""
import cProfile
import io
import os
import pstats
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    from splatloc_amd.camera import PinholeCamera
    from splatloc_amd.optim import Adam as FusedAdam
    from splatloc_amd.synthetic import WORKLOADS, make_workload
    from splatloc_amd.training import color_refinement_step, map_step
    kind = sys.argv[1] if len(sys.argv) > 1 else "refine"
    workload = sys.argv[2] if len(sys.argv) > 2 else "S0"
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    dev = torch.device("cuda:0")
    wl = WORKLOADS[workload]
    sc = make_workload(workload)
    P0, W, H = wl["P"], wl["W"], wl["H"]
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
        _kp_score=par(torch.rand(P0, 1, generator=g)), _scaling=par(torch.log(sc.scales)), _rotation=par(sc.rotations.clone()),
        active_sh_degree=0, max_sh_degree=0, lr_init=1.6e-4 * 6.0, lr_final=1.6e-6 * 6.0, lr_delay_mult=0.01, max_steps=30000,
        percent_dense=0.01, primitive_reg=True)
    pc.optimizer = FusedAdam([{"params": [getattr(pc, attr[k])], "lr": lr[k], "name": k} for k in names], lr=0.0, eps=1e-15)
    pc.max_radii2D = torch.zeros(P0, device=dev)
    pc.xyz_gradient_accum = torch.zeros(P0, 1, device=dev)
    pc.denom = torch.zeros(P0, 1, device=dev)
    views = []
    for k in range(8):
        ang = torch.tensor(0.02 * (k - 4))
        R = torch.tensor([[torch.cos(ang), 0, torch.sin(ang)], [0, 1, 0], [-torch.sin(ang), 0, torch.cos(ang)]])
        cam = PinholeCamera(W, H, W / 2.0, W / 2.0, (W - 1) / 2.0, (H - 1) / 2.0, R, torch.tensor([0.01 * k, 0.0, 0.0])).to(dev)
        cam.original_image = torch.rand(3, H, W, generator=g).to(dev)
        cam.depth = (0.5 + 3 * torch.rand(H, W, generator=g)).to(dev)
        cam.kp_score = (torch.rand(H, W, generator=g) ** 4).to(dev)
        cam.exposure_a = torch.zeros(1, device=dev, requires_grad=True)
        cam.exposure_b = torch.zeros(1, device=dev, requires_grad=True)
        views.append(cam)
    bg = torch.zeros(3, device=dev)
    pipe = types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False)
    cfg = {"Training": {"rgb_boundary_threshold": 0.01, "primitive_reg": True}}
    it = [0]

    def loop(n):
        for _ in range(n):
            it[0] += 1
            if kind == "refine":
                color_refinement_step(views[it[0] % 8], pc, pipe, bg, 0.2, it[0])
            else:
                map_step([views[(it[0] + j) % 8] for j in range(5)], pc, pipe, bg, cfg, it[0])

    loop(30)
    torch.cuda.synchronize(dev)
    pr = cProfile.Profile()
    pr.enable()
    loop(N)
    pr.disable()
    torch.cuda.synchronize(dev)
    for key in ("cumulative", "tottime"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(38)
        txt = s.getvalue()
        print(f"==== {kind} {workload}: per-call microseconds = column x 1e6 / {N}; sorted by {key}")
        print("\n".join(l[:150] for l in txt.splitlines()[4:50]))


if __name__ == "__main__":
    main()
