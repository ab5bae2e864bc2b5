#!/usr/bin/env python3
"""Replica consistency of the frame-parallel map step: run under
`python -m torch.distributed.run --nproc-per-node N tools/replica_check.py` (backend from SPLATLOC_DIST_BACKEND, default
nccl = RCCL; the GPU test drives it with gloo and both ranks on one GPU).  Every rank holds a replica of one seeded
scene, the 5 views of each window are dealt to the ranks, and `splatloc_amd.training.map_step` runs `--steps`
iterations including one densify_and_prune and one opacity reset.  Afterwards every parameter tensor, every Adam moment,
the step counters, the statistics and the row count must be BIT-IDENTICAL on all ranks (sha256 of the bytes): the
replicas never exchange parameters, only reduced gradients / statistics, so any divergence would grow silently.
Prints one JSON line on rank 0; exit code 1 on a mismatch."""
import argparse
import hashlib
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def make_replica(dev, P=20000, W=320, H=240):
    """One seeded replica of a small SplatLoc model (the reference's parameter groups, fused Adam) + 10 views with ground-truth
    images — what every rank of a frame-parallel run holds.  Returns (pc, views, bg, pipe, cfg, dens)."""
    from splatloc_amd.camera import PinholeCamera
    from splatloc_amd.optim import Adam
    from splatloc_amd.synthetic import make_scene
    sc = make_scene(P, W, H, 4, seed=77, scale_median=0.03)
    g = torch.Generator().manual_seed(5)
    names = ("xyz", "f_dc", "f_rest", "opacity", "marker", "kp_score", "scaling", "rotation")
    attr = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "marker": "_marker",
            "kp_score": "_kp_score", "scaling": "_scaling", "rotation": "_rotation"}
    lr = {"xyz": 1.6e-4 * 6.0, "f_dc": 2.5e-3, "f_rest": 2.5e-3 / 20, "opacity": 5e-2, "marker": 5e-2, "kp_score": 5e-2,
          "scaling": 1e-3 * 6.0, "rotation": 1e-3}
    par = lambda t: torch.nn.Parameter(t.to(dev).contiguous().requires_grad_(True))  # noqa: E731
    pc = types.SimpleNamespace(
        _xyz=par(sc.means3D.clone()), _features_dc=par(((sc.features[:, :3] - 0.5) / 0.28209479177387814)[:, None, :].contiguous()),
        _features_rest=par(torch.zeros(P, 0, 3)), _opacity=par(torch.logit(sc.opacities.clamp(1e-4, 1 - 1e-4))),
        _marker=par((torch.rand(P, 1, generator=g) < 0.05).float() * torch.rand(P, 1, generator=g) * 0.9),
        _kp_score=par(torch.rand(P, 1, generator=g)), _scaling=par(torch.log(sc.scales)), _rotation=par(sc.rotations.clone()),
        active_sh_degree=0, max_sh_degree=0, percent_dense=0.01, primitive_reg=True, lr_init=1.6e-4 * 6.0,
        lr_final=1.6e-6 * 6.0, lr_delay_mult=0.01, max_steps=30000)
    pc.optimizer = Adam([{"params": [getattr(pc, attr[k])], "lr": lr[k], "name": k} for k in names], lr=0.0, eps=1e-15)
    pc.xyz_gradient_accum = torch.zeros(P, 1, device=dev)
    pc.denom = torch.zeros(P, 1, device=dev)
    pc.max_radii2D = torch.zeros(P, device=dev)
    views = []
    for k in range(10):
        ang = torch.tensor(0.03 * (k - 5))
        R = torch.tensor([[torch.cos(ang), 0, torch.sin(ang)], [0, 1, 0], [-torch.sin(ang), 0, torch.cos(ang)]])
        cam = PinholeCamera(W, H, W / 2.0, W / 2.0, (W - 1) / 2.0, (H - 1) / 2.0, R, torch.tensor([0.02 * k, 0.0, 0.0])).to(dev)
        cam.original_image = torch.rand(3, H, W, generator=g).to(dev)
        cam.depth = (0.5 + 3 * torch.rand(H, W, generator=g)).to(dev)
        cam.kp_score = (torch.rand(H, W, generator=g) ** 4).to(dev)
        cam.exposure_a = torch.zeros(1, device=dev, requires_grad=True)
        cam.exposure_b = torch.zeros(1, device=dev, requires_grad=True)
        views.append(cam)
    bg = torch.zeros(3, device=dev)
    pipe = types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False)
    cfg = {"Training": {"rgb_boundary_threshold": 0.01, "primitive_reg": True}}
    dens = dict(grad_threshold=0.0002, min_opacity=0.005, extent=6.0, size_threshold=20, every=3, offset=2)
    return pc, views, bg, pipe, cfg, dens


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--P", type=int, default=20000)
    ap.add_argument("--window", type=int, default=5, help="views per window (1: every rank but the first has no work)")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    dev = torch.device("cuda", local % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SPLATLOC_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    from splatloc_amd import training
    from splatloc_amd.training import map_step
    pc, views, bg, pipe, cfg, dens = make_replica(dev, args.P)
    rows, colls = [], []
    for it in range(1, args.steps + 1):
        perm = torch.randperm(len(views), generator=torch.Generator().manual_seed(1000 + it))[:args.window]   # the same draw on every rank
        map_step([views[i] for i in perm], pc, pipe, bg, cfg, it, densify=dens, gaussian_reset=4, seed=9)
        rows.append(int(pc._xyz.shape[0]))
        colls.append(training.LAST_STEP_INFO.get("collectives"))
    torch.cuda.synchronize(dev)
    digest = {}
    h = lambda t: hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()  # noqa: E731
    for grp in pc.optimizer.param_groups:
        p = grp["params"][0]
        digest["param_" + grp["name"]] = h(p)
        st = pc.optimizer.state.get(p, None)
        if st:
            digest["m_" + grp["name"]], digest["v_" + grp["name"]] = h(st["exp_avg"]), h(st["exp_avg_sq"])
            digest["step_" + grp["name"]] = float(st["step"])
        digest["lr_" + grp["name"]] = grp["lr"]
    for k in ("xyz_gradient_accum", "denom", "max_radii2D"):
        digest[k] = h(getattr(pc, k))
    digest["rows"] = rows
    ok = True
    if world > 1:
        all_d = [None] * world
        dist.all_gather_object(all_d, digest)
        bad = [k for k in digest if any(d[k] != all_d[0][k] for d in all_d)]
        ok = not bad
    else:
        bad = []
    if rank == 0:
        print(json.dumps({"world": world, "steps": args.steps, "rows_per_step": rows, "identical": ok, "mismatched": bad,
                          "tensors_compared": len(digest), "window": args.window,
                          "collectives_per_step": (max(c for c in colls if c is not None) if world > 1 else 0)}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
