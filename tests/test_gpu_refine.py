"""splatloc_amd.training.color_refinement_step against tests/golden/refine_step.npz — three consecutive iterations of
SplatLoc.color_refinement (train_gaussians.py:272-297) recorded from the reference's own render / l1_loss / ssim /
GaussianModel / torch.optim.Adam with the CPU oracle standing in for the un-vendored rasterizer — needs an MI355X.

Every iteration starts from the RECORDED state of the previous one (parameters, Adam moments and step counters,
max_radii2D, the xyz learning rate), runs ONE step on the device and is compared with the recording: loss, every
gradient as it reaches the optimizer (xyz after the key-primitive gate; `_marker.grad is None`; the kp_score column
gets a gradient of zeros, hence Adam state, as in the reference), the new parameters, moments, max_radii2D and the
learning rate.  The device sums gradients with float atomics, and with eps = 1e-15 Adam's first steps move a
parameter by lr * sign(g): an element whose gradient is rounding noise around zero may step the other way — such
elements are bounded in number (1e-3) and in size (2 lr)."""
import os
import types

import numpy as np
import pytest
import torch

from tests.helpers import assert_grad_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "marker", "kp_score", "scaling", "rotation")
ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
        "marker": "_marker", "kp_score": "_kp_score", "scaling": "_scaling", "rotation": "_rotation"}


def _load_state(d, pre, adam_cls, dev):
    gm = types.SimpleNamespace(active_sh_degree=0, max_sh_degree=0, lr_init=0.0016 * 6.0, lr_final=0.0000016 * 6.0,
                               lr_delay_mult=0.01, max_steps=30000)
    par = lambda a: torch.nn.Parameter(torch.from_numpy(a).to(dev).contiguous().requires_grad_(True))  # noqa: E731
    for k in GROUPS:
        setattr(gm, ATTR[k], par(d[pre + k]))
    gm.optimizer = adam_cls([{"params": [getattr(gm, ATTR[k])], "lr": float(d[f"{pre}lr_{k}"]), "name": k} for k in GROUPS],
                            lr=0.0, eps=1e-15)
    for grp in gm.optimizer.param_groups:
        k = grp["name"]
        if bool(d[f"{pre}has_state_{k}"]):
            gm.optimizer.state[grp["params"][0]] = {
                "step": torch.tensor(float(d[f"{pre}step_{k}"])), "exp_avg": torch.from_numpy(d[f"{pre}m_{k}"]).to(dev),
                "exp_avg_sq": torch.from_numpy(d[f"{pre}v_{k}"]).to(dev)}
    gm.max_radii2D = torch.from_numpy(d[pre + "max_radii"]).to(dev)
    return gm


def _close_but_sign_flips(name, got, ref, lr, rtol, atol):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    bad = np.abs(got - ref) > rtol * np.abs(ref) + atol
    assert bad.mean() <= 1e-3, f"{name}: {bad.sum()} of {bad.size} elements off"
    assert np.abs(got - ref).max() <= 2.05 * lr + rtol * np.abs(ref).max() + atol, f"{name}: worst {np.abs(got - ref).max():.3e} (lr {lr:.3e})"


@pytest.mark.parametrize("adam", ["torch", "fused"])
def test_color_refinement_steps_match_reference_recording(golden_dir, adam):
    from splatloc_amd.camera import PinholeCamera
    from splatloc_amd.optim import Adam as FusedAdam
    from splatloc_amd.training import color_refinement_step
    d = np.load(os.path.join(golden_dir, "refine_step.npz"))
    dev = torch.device(DEV)
    fx, fy, cx, cy, W, H = (float(v) for v in d["intr"][:6])
    W, H = int(W), int(H)
    pipe = types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False)
    bg = torch.zeros(3, device=dev)
    lam = float(d["lambda_dssim"])
    adam_cls = torch.optim.Adam if adam == "torch" else FusedAdam
    for it in (1, 2, 3):
        gm = _load_state(d, f"s{it - 1}_", adam_cls, dev)
        T = torch.from_numpy(d[f"view{it - 1}_T"])
        cam = PinholeCamera(W, H, fx, fy, cx, cy, T[:3, :3], T[:3, 3]).to(dev)
        cam.original_image = torch.from_numpy(d[f"view{it - 1}_color"]).to(dev)
        seen = {}
        real_step = gm.optimizer.step

        def spy(*a, _gm=gm, **kw):
            for k in GROUPS:
                g = getattr(_gm, ATTR[k]).grad
                seen[k] = None if g is None else g.detach().clone()
            return real_step(*a, **kw)

        gm.optimizer.step = spy
        loss = color_refinement_step(cam, gm, pipe, bg, lam, it, primitive_reg=True)
        torch.cuda.synchronize()
        pre = f"it{it}_"
        np.testing.assert_allclose(float(loss), float(d[pre + "loss"]), rtol=2e-5)
        # gradients as they reach the optimizer
        for k in GROUPS:
            has = bool(d[pre + "has_grad_" + k])
            assert (seen[k] is not None) == has, f"{k}: gradient presence differs from the reference"
            if not has or not seen[k].numel():
                continue
            g = seen[k].cpu().numpy()
            if k == "xyz" and adam == "fused":
                # the gate is applied inside the fused Adam launch, not to .grad: apply it here for the comparison
                g = g * (d[f"s{it - 1}_marker"] <= 0.005)
            assert_grad_close(f"it{it} grad {k}", g, d[pre + "grad_" + k], rtol=3e-3, atol_scale=2e-4)
        assert float(np.abs(seen["kp_score"].cpu().numpy()).max()) == 0.0      # in the graph, not in the loss
        # the step
        post = f"s{it}_"
        for grp in gm.optimizer.param_groups:
            k = grp["name"]
            p = grp["params"][0]
            lr = float(d[f"s{it - 1}_lr_{k}"])
            if not p.numel():
                continue
            _close_but_sign_flips(f"it{it} param {k}", p.detach().cpu().numpy(), d[post + k], lr, rtol=1e-6, atol=0.02 * lr + 1e-9)
            st = gm.optimizer.state.get(p, None)
            assert bool(d[f"{post}has_state_{k}"]) == bool(st is not None and len(st)), k
            if st is not None and len(st):
                assert float(st["step"]) == float(d[f"{post}step_{k}"])
                assert_grad_close(f"it{it} exp_avg {k}", st["exp_avg"].cpu().numpy(), d[f"{post}m_{k}"], rtol=3e-3, atol_scale=2e-4)
                assert_grad_close(f"it{it} exp_avg_sq {k}", st["exp_avg_sq"].cpu().numpy(), d[f"{post}v_{k}"], rtol=6e-3, atol_scale=2e-4)
            np.testing.assert_allclose(grp["lr"], float(d[f"{post}lr_{k}"]), rtol=1e-12)
        assert np.array_equal(gm.max_radii2D.cpu().numpy(), d[post + "max_radii"])
        assert np.array_equal(d[pre + "radii"], d[pre + "radii"].astype(np.int32))


@pytest.mark.parametrize("adam", ["torch", "fused"])
def test_map_step_function_matches_reference_iterations(golden_dir, adam):
    """splatloc_amd.training.map_step against tests/golden/map_iteration.npz: three COMPLETE iterations of the loop body of
    SplatLoc.map (train_gaussians.py:188-267) recorded from the reference's own code — 5 views drawn from 7, losses,
    regulariser, backward, key gate, statistics per view, the opacity reset of iteration 2, Adam, lr schedule.  Each
    iteration starts from the recorded state of the previous one and is compared with the recording."""
    from splatloc_amd.camera import PinholeCamera
    from splatloc_amd.optim import Adam as FusedAdam
    from splatloc_amd.training import map_step
    d = np.load(os.path.join(golden_dir, "map_iteration.npz"))
    dev = torch.device(DEV)
    fx, fy, cx, cy, W, H = (float(v) for v in d["intr"][:6])
    W, H = int(W), int(H)
    pipe = types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False)
    cfg = {"Training": {"primitive_reg": True, "rgb_boundary_threshold": 0.01}}
    bg = torch.zeros(3, device=dev)
    cams = []
    for k in range(7):
        T = torch.from_numpy(d[f"view{k}_T"])
        cam = PinholeCamera(W, H, fx, fy, cx, cy, T[:3, :3], T[:3, 3]).to(dev)
        cam.original_image = torch.from_numpy(d[f"view{k}_color"]).to(dev)
        cam.depth = d[f"view{k}_depth"]
        cam.kp_score = torch.from_numpy(d[f"view{k}_kp"]).to(dev)
        cam.exposure_a = torch.zeros(1, device=dev, requires_grad=True)
        cam.exposure_b = torch.zeros(1, device=dev, requires_grad=True)
        cams.append(cam)
    adam_cls = torch.optim.Adam if adam == "torch" else FusedAdam
    for it in (1, 2, 3):
        gm = _load_state(d, f"s{it - 1}_", adam_cls, dev)
        gm.percent_dense, gm.primitive_reg = 0.01, True
        gm.xyz_gradient_accum = torch.from_numpy(d[f"s{it - 1}_accum"]).to(dev)
        gm.denom = torch.from_numpy(d[f"s{it - 1}_denom"]).to(dev)
        loss = map_step([cams[i] for i in d[f"it{it}_views"]], gm, pipe, bg, cfg, it, densify=None,
                        gaussian_reset=int(d["gaussian_reset"]))
        torch.cuda.synchronize()
        np.testing.assert_allclose(float(loss), float(d[f"it{it}_loss"]), rtol=2e-5)
        post = f"s{it}_"
        for grp in gm.optimizer.param_groups:
            k = grp["name"]
            p = grp["params"][0]
            lr = float(d[f"s{it - 1}_lr_{k}"])
            if not p.numel():
                continue
            _close_but_sign_flips(f"it{it} param {k}", p.detach().cpu().numpy(), d[post + k], lr, rtol=1e-6, atol=0.02 * lr + 1e-9)
            st = gm.optimizer.state.get(p, None)
            assert bool(d[f"{post}has_state_{k}"]) == bool(st is not None and len(st)), k
            if st is not None and len(st):
                assert float(st["step"]) == float(d[f"{post}step_{k}"]), k
                assert_grad_close(f"it{it} exp_avg {k}", st["exp_avg"].cpu().numpy(), d[f"{post}m_{k}"], rtol=3e-3, atol_scale=2e-4)
                assert_grad_close(f"it{it} exp_avg_sq {k}", st["exp_avg_sq"].cpu().numpy(), d[f"{post}v_{k}"], rtol=6e-3, atol_scale=2e-4)
            np.testing.assert_allclose(grp["lr"], float(d[f"{post}lr_{k}"]), rtol=1e-12)
        assert np.array_equal(gm.max_radii2D.cpu().numpy(), d[post + "max_radii"])
        assert np.array_equal(gm.denom.cpu().numpy(), d[post + "denom"])
        assert_grad_close(f"it{it} xyz_gradient_accum", gm.xyz_gradient_accum.cpu().numpy(), d[post + "accum"], rtol=3e-3, atol_scale=2e-4)


@pytest.mark.parametrize("workload", ["S0", "S2-ref-layout"])
def test_raw_parameter_backward_is_the_two_kernel_chain_bit_for_bit(workload):
    """`color_refinement_step`'s graph-free path runs the parameter activations INSIDE the rasterizer's per-Gaussian kernels: the
    projection kernel reads the raw tensors (splatraster_forward_window_geometry_raw) and the backward writes the RAW parameters'
    gradients itself (splatraster_backward_window_raw: the chain through exp / normalize / sigmoid / SH degree 0 + clamp and the sum
    of the accumulator rows' colour columns) instead of activate_forward + ... + gather_dcolors + preprocess_bwd + activate_backward.  Same
    arithmetic from one shared header (csrc/activation_math.h): with the deterministic-sum compositing the gradients of every
    parameter group must be IDENTICAL bit for bit, at 10k and at 500k Gaussians (full reference layout)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import map_idle
    from splatloc_amd import _native, training
    dev = torch.device(DEV)
    pc, views = map_idle.build(workload, dev)
    bg = torch.zeros(3, device=dev)
    grads = {}
    _native.set_deterministic(True)
    try:
        for raw in (False, True):
            training.RAW_BACKWARD = raw
            for p in (pc._xyz, pc._features_dc, pc._features_rest, pc._opacity, pc._kp_score, pc._scaling, pc._rotation):
                p.grad = None
            captured = {}
            real_step = pc.optimizer.step

            def capture():      # the gradients as they reach the optimizer
                for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_kp_score", "_scaling", "_rotation"):
                    g = getattr(pc, k).grad
                    captured[k] = None if g is None else g.detach().clone()
            pc.optimizer.step = capture
            try:
                loss = training._color_refinement_step_direct(views[1], pc, bg, 0.2, 7, True)
            finally:
                pc.optimizer.step = real_step
            torch.cuda.synchronize()
            captured["loss"] = loss.detach().reshape(1).clone()      # (the forward: activations inside the projection kernel vs in front of it)
            grads[raw] = captured
            assert training._raw_backward_ok(pc) == raw
    finally:
        training.RAW_BACKWARD = True
        _native.set_deterministic(False)
    for k, a in grads[False].items():
        b = grads[True][k]
        assert (a is None) == (b is None), k
        if a is None:
            continue
        assert a.shape == b.shape and torch.isfinite(b).all(), k
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (k, float((a - b).abs().max()))
    assert float(grads[True]["_scaling"].abs().max()) > 0 and float(grads[True]["_features_dc"].abs().max()) > 0
    assert float(grads[True]["_kp_score"].abs().max()) == 0.0       # the refinement loss reaches the RGB channels only


@pytest.mark.parametrize("workload", ["S0", "S2-ref-layout"])
def test_map_step_raw_parameter_path_is_the_two_kernel_chain_bit_for_bit(workload):
    """The same for the graph-free MAP step (five views, the isotropic regulariser's term joining dL/dscales inside the per-Gaussian
    backward before the chain through exp): gradients of every parameter group, the per-view dL/dmeans2D and the loss identical bit for
    bit to activate_forward + window + regulariser + activate_backward, with the deterministic-sum compositing."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import map_idle
    import types as _types
    from splatloc_amd import _native, training
    dev = torch.device(DEV)
    pc, views = map_idle.build(workload, dev)
    bg = torch.zeros(3, device=dev)
    pipe = _types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False)
    cfg = {"Training": {"rgb_boundary_threshold": 0.01, "primitive_reg": True}}
    keys = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_kp_score", "_scaling", "_rotation")
    res = {}
    _native.set_deterministic(True)
    try:
        for raw in (False, True):
            training.RAW_BACKWARD = raw
            for k in keys:
                getattr(pc, k).grad = None
            for cam in views:
                cam.exposure_a.grad = cam.exposure_b.grad = None
            out = training._map_grads_direct(views, pc, pipe, bg, cfg, True)
            assert out is not None
            pkgs, loss, grads2d = out
            torch.cuda.synchronize()
            res[raw] = ({k: (None if getattr(pc, k).grad is None else getattr(pc, k).grad.detach().clone()) for k in keys},
                        [g.detach().clone() for g in grads2d], loss.detach().reshape(1).clone(), [p["render"].detach().clone() for p in pkgs])
    finally:
        training.RAW_BACKWARD = True
        _native.set_deterministic(False)
    (ga, m2a, la, ima), (gb, m2b, lb, imb) = res[False], res[True]
    assert torch.equal(la.view(torch.int32), lb.view(torch.int32))
    for a, b in zip(ima, imb):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    for a, b in zip(m2a, m2b):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    for k in keys:
        a, b = ga[k], gb[k]
        assert (a is None) == (b is None), k
        if a is not None:
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (k, float((a - b).abs().max()))
    assert float(gb["_scaling"].abs().max()) > 0 and float(gb["_kp_score"].abs().max()) > 0
