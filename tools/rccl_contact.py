#!/usr/bin/env python3
"""First contact with RCCL on ONE GPU (/root/reference/replica.sh:1-6 is what the 8-GPU
configuration replaces).  Run as a fresh process: `python tools/rccl_contact.py`.

Creates a WORLD-SIZE-1 process group with backend "nccl" (= RCCL on ROCm) and `device_id=`, and issues on it — with the
one-rank early returns of `splatloc_amd.frame_parallel` bypassed by `force=True` — every collective call the N > 1 path
makes, on the very allocations it makes them on:

  1. the in-place span SUM on the gradient allocation a real `rasterize_window(..., grad_span=[])` backward produced
     (autograd owns it), with the [2, P] statistics tail, + the MAX over max_radii2D + the 32-byte length header;
  2. the same exchange as the aliased `reduce_scatter_tensor` + `all_gather_into_tensor` pair (`mode="rs_ag"`: gloo emulates
     it with an all-reduce, so before this script that branch had never executed anywhere);
  3. the packed path (`torch.cat` + views of the reduced buffer) in both modes, with a length that is not a multiple of 1..8;
  4. `allreduce_grads` (span + bucket paths) and `sync_densification_stats` (the round-1..3 building blocks);
  5. `broadcast_model` of a model with Adam state;
  6. two `training.map_step` iterations (densify on the 2nd) with SPLATLOC_FORCE_COLLECTIVES semantics;
  7. `barrier`, `all_gather_object`, `destroy_process_group`.

A SUM / MAX / broadcast over one rank must leave every value bit-identical: asserted.  Prints ONE JSON line (backend, RCCL
version, the env RCCL depends on, what ran) and exits 0; any failure is a non-zero exit with the traceback.
What this CANNOT cover: peer access over xGMI, more than one communicator rank, ring / tree selection."""
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
if "MASTER_PORT" not in os.environ:
    with socket.socket() as _s:
        _s.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(_s.getsockname()[1])

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    t_start = time.perf_counter()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    ran = {}
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1

    from splatloc_amd import GaussianRasterizationSettings, rasterize_window
    from splatloc_amd import frame_parallel as fp
    from splatloc_amd import training
    from splatloc_amd.camera import PinholeCamera
    from splatloc_amd.densify import add_densification_stats_window
    from splatloc_amd.synthetic import make_scene
    import replica_check

    # without force a group of one exchanges nothing
    g0, e0, info0 = fp.reduce_step([torch.ones(4, device=dev)], max_extras=[torch.ones(4, device=dev)])
    assert info0["collectives"] == 0 and not fp.collectives_active()

    # ---- 1 + 2: the window backward's own allocation, ring and rs_ag
    P, W, H, C = 6000, 320, 240, 4
    sc = make_scene(P, W, H, C, seed=3, scale_median=0.03).to(dev)
    leaf = lambda t: t.clone().requires_grad_(True)  # noqa: E731
    params = [leaf(sc.means3D), leaf(sc.features), leaf(sc.opacities), leaf(sc.scales), leaf(sc.rotations)]
    settings = []
    for j in range(3):
        ang = torch.tensor(0.02 * (j - 1))
        Rm = torch.tensor([[torch.cos(ang), 0, torch.sin(ang)], [0, 1, 0], [-torch.sin(ang), 0, torch.cos(ang)]])
        cam = PinholeCamera(W, H, W / 2.0, W / 2.0, (W - 1) / 2.0, (H - 1) / 2.0, Rm, torch.tensor([0.01 * j, 0.0, 0.0])).to(dev)
        settings.append(GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, sc.bg, 1.0, cam.world_view_transform,
                                                      cam.full_proj_transform, 0, cam.camera_center, False, False))
    for mode in ("ring", "rs_ag"):
        for p in params:
            p.grad = None
        block = torch.zeros((3, P, 3), device=dev)
        carriers = [block[k].requires_grad_(True) for k in range(3)]
        span = []
        outs = rasterize_window(settings, params[0], carriers, params[1], params[2], scales=params[3], rotations=params[4],
                                grad_span=span)
        torch.autograd.backward([t for o in outs for t in o[:3]], [g for _ in range(3) for g in (sc.dL_dcolor, sc.dL_ddepth, sc.dL_dalpha)])
        assert len(span) == 1
        inc = span[0]["tail"]
        max_radii = torch.zeros(P, device=dev)
        add_densification_stats_window([m.grad for m in carriers], [o[3] for o in outs], inc[0], inc[1], max_radii)
        torch.cuda.synchronize(dev)
        before = [p.grad.clone() for p in params] + [inc.clone(), max_radii.clone()]
        ptrs = [p.grad.data_ptr() for p in params]
        g_out, inc_out, info = fp.reduce_step([p.grad for p in params], sum_extras=[inc[0], inc[1]], max_extras=[max_radii],
                                              mode=mode, force=True)
        torch.cuda.synchronize(dev)
        assert info["sum_path"] == "in-place span", info
        assert info["collectives"] == (3 if mode == "rs_ag" else 2) and info["header_collectives"] == 1, info
        assert [g.data_ptr() for g in g_out] == ptrs                   # reduced where the backward left them
        after = [p.grad for p in params] + [inc, max_radii]
        for a, b in zip(before, after):
            assert torch.equal(a, b) and torch.isfinite(b).all()
        assert float(before[0].abs().sum()) > 0 and float(max_radii.max()) > 0
        ran[f"window_span_{mode}"] = {k: info[k] for k in ("collectives", "sum_path", "sum_bytes", "max_bytes", "header_ms")}

    # ---- 3: packed path, both modes (length 1009 + 7: not a multiple of anything useful)
    for mode in ("ring", "rs_ag"):
        a = torch.randn(1009, device=dev)
        b = torch.randn(7, 1, device=dev)
        m1, m2 = torch.rand(33, device=dev), torch.rand(5, device=dev)
        ref = [t.clone() for t in (a, b, m1, m2)]
        g_out, e_out, info = fp.reduce_step([a], sum_extras=[b], max_extras=[m1, m2], mode=mode, force=True)
        torch.cuda.synchronize(dev)
        assert info["sum_path"] == "packed", info
        assert torch.equal(g_out[0], ref[0]) and torch.equal(e_out[0], ref[1]) and torch.equal(m1, ref[2]) and torch.equal(m2, ref[3])
        ran[f"packed_{mode}"] = info["collectives"]

    # ---- 4: the building blocks
    flat = torch.randn(100, device=dev)
    u, v, w = flat[0:30].view(10, 3), flat[32:42].view(10, 1), flat[44:84].view(10, 4)
    far = torch.randn(50, 3, device=dev)[:, :2]        # non-contiguous: bucket path with a staging cat
    ref = [t.clone() for t in (u, v, w, far)]
    path = fp.allreduce_grads([u, None, v, w, far], force=True)
    torch.cuda.synchronize(dev)
    assert path["spans"] == 1 and path["buckets"] == 1, path
    for a, b in zip(ref, (u, v, w, far)):
        assert torch.equal(a, b)
    acc, den, rad = torch.rand(P, 1, device=dev), torch.ones(P, 1, device=dev), torch.rand(P, device=dev)
    ref = [t.clone() for t in (acc, den, rad)]
    fp.sync_densification_stats(acc, den, rad, force=True)
    torch.cuda.synchronize(dev)
    for a, b in zip(ref, (acc, den, rad)):
        assert torch.equal(a, b)
    ran["allreduce_grads"] = path

    # ---- 6 (first, so that the model has Adam state) + 5: map steps with forced collectives, then the broadcast
    pc, views, bg, pipe, cfg, dens = replica_check.make_replica(dev, 8000)
    fp.FORCE_COLLECTIVES = True
    steps = []
    for it in (1, 2):
        training.REDUCE_MODE = "ring" if it == 1 else "rs_ag"
        perm = torch.randperm(len(views), generator=torch.Generator().manual_seed(1000 + it))[:5]
        loss = training.map_step([views[i] for i in perm], pc, pipe, bg, cfg, it, densify=dens, gaussian_reset=4, seed=9)
        assert torch.isfinite(loss).all()
        steps.append({k: training.LAST_STEP_INFO.get(k) for k in ("collectives", "sum_path", "mode", "render_path", "header_ms")})
        assert training.LAST_STEP_INFO["collectives"] == (2 if it == 1 else 3), training.LAST_STEP_INFO
    fp.FORCE_COLLECTIVES = False
    training.REDUCE_MODE = "ring"
    ran["map_steps"] = steps
    ran["rows_after_densify"] = int(pc._xyz.shape[0])
    state = []
    for grp in pc.optimizer.param_groups:
        p = grp["params"][0]
        state.append(p.data.clone())
        st = pc.optimizer.state.get(p)
        if st and "exp_avg" in st:
            state += [st["exp_avg"].clone(), st["exp_avg_sq"].clone()]
    lrs = [grp["lr"] for grp in pc.optimizer.param_groups]
    nbytes = fp.broadcast_model(pc, src=0, force=True)
    torch.cuda.synchronize(dev)
    assert nbytes > 0
    k = 0
    for grp in pc.optimizer.param_groups:
        p = grp["params"][0]
        assert torch.equal(state[k], p.data)
        k += 1
        st = pc.optimizer.state.get(p)
        if st and "exp_avg" in st:
            assert torch.equal(state[k], st["exp_avg"]) and torch.equal(state[k + 1], st["exp_avg_sq"])
            k += 2
    assert lrs == [grp["lr"] for grp in pc.optimizer.param_groups]
    ran["broadcast_model_bytes"] = nbytes

    # ---- 7
    dist.barrier()
    gathered = [None]
    dist.all_gather_object(gathered, {"rank": 0})
    assert gathered == [{"rank": 0}]
    try:
        rccl = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception as e:  # noqa: BLE001
        rccl = f"unavailable ({type(e).__name__})"
    out = {"ok": True, "backend": dist.get_backend(), "world_size": dist.get_world_size(), "rccl_version": rccl,
           "torch": torch.__version__, "hip": getattr(torch.version, "hip", None), "device": torch.cuda.get_device_name(dev),
           "env": {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG", "MASTER_ADDR")},
           "ran": ran, "seconds": round(time.perf_counter() - t_start, 2),
           "not_covered": "peer access over xGMI, > 1 communicator rank, RCCL's ring / tree / direct algorithm selection"}
    dist.destroy_process_group()
    assert not dist.is_initialized()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
