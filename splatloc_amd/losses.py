"""SplatLoc's per-view mapping loss as ONE HIP pass that also produces the gradient w.r.t. the
rendered buffers.

`mapping_loss(config, image, depth, marker, viewpoint, initialization=False)` equals

    get_loss_mapping(config, image, depth, viewpoint, opacity, initialization)      # utils/utils.py:55-82
    + get_loss_marker(config, marker, viewpoint.kp_score)                           # train_gaussians.py:38-42

(the per-view sum of train_gaussians.py:217-218) in value and in the gradients that reach
`image`, `depth`, `marker`, `viewpoint.exposure_a`, `viewpoint.exposure_b`
(tests/test_gpu_losses.py, against the fixture recorded from the reference's autograd).
No CPU fallback: tensors must be on the ROCm device.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _native
from .rasterizer import _prep, _ptr, _require_gpu, _stream, _on_device


def _mapping_loss_launch(image, depth, marker, gt_image, gt_depth, kp, threshold: float, exposure, out=None):
    """One launch of the fused mapping loss: returns (g_image, g_depth, g_marker, out) with out = [loss_mapping, loss_marker,
    dL/dexposure_a, dL/dexposure_b].  `out`: optional preallocated 4-float tensor (a row of a per-window table)."""
    lib = _native.load()
    _require_gpu(image, "image")
    dev = image.device
    H, W = int(image.shape[-2]), int(image.shape[-1])
    HW = H * W
    if tuple(image.shape) != (3, H, W) or depth.numel() != HW or marker.numel() != HW:
        raise RuntimeError("mapping_loss: expected image [3,H,W], depth [1,H,W], marker [H,W]")
    im, de, ma, gi, gd = (_prep(t, dev) for t in (image, depth, marker, gt_image, gt_depth))
    # the BCE target is the score map itself (`gt.view(-1).float()`, train_gaussians.py:40): a soft
    # target in [0, 1] on real data (utils/dataset.py:94), 0/1 when a bool mask is passed
    k8 = _prep(kp.to(torch.float32), dev)
    ex = _prep(exposure, dev) if exposure is not None else None
    f32 = dict(dtype=torch.float32, device=dev)
    g_image = torch.empty((3, H, W), **f32)
    g_depth = torch.empty(tuple(depth.shape), **f32)
    g_marker = torch.empty(tuple(marker.shape), **f32)
    if out is None:
        out = torch.empty((4,), **f32)
    ws = torch.empty((lib.splatraster_mapping_loss_workspace_bytes(HW),), dtype=torch.uint8, device=dev)
    with _on_device(dev):
        _native.check(lib.splatraster_mapping_loss(
            HW, _ptr(im), _ptr(de), _ptr(ma), _ptr(gi), _ptr(gd), _ptr(k8), C.c_float(float(threshold)), _ptr(ex),
            _ptr(g_image), _ptr(g_depth), _ptr(g_marker), _ptr(out), _ptr(ws), _stream(dev)), "mapping_loss")
    return g_image, g_depth, g_marker, out


class _MappingLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, depth, marker, gt_image, gt_depth, kp, threshold: float, exposure):
        g_image, g_depth, g_marker, out = _mapping_loss_launch(image, depth, marker, gt_image, gt_depth, kp, threshold, exposure)
        ctx.save_for_backward(g_image, g_depth, g_marker, out)
        ctx.has_exposure = exposure is not None
        return out[0] + out[1]

    @staticmethod
    def backward(ctx, g):
        g_image, g_depth, g_marker, out = ctx.saved_tensors
        g_exp = g * out[2:4] if ctx.has_exposure else None
        return g * g_image, g * g_depth, g * g_marker, None, None, None, None, g_exp


def mapping_loss_tensors(image, depth, marker, gt_image, gt_depth, kp, rgb_boundary_threshold, exposure_a=None,
                         exposure_b=None):
    """Tensor-level entry: image [3,H,W], depth [1,H,W], marker [H,W] (logits), gt_image [3,H,W],
    gt_depth [H,W], kp [H,W] float score map in [0,1] (or bool); exposure_a / exposure_b one-element tensors or None (no affine)."""
    exposure = None
    if exposure_a is not None:
        exposure = torch.cat((exposure_a.reshape(1), exposure_b.reshape(1))).to(torch.float32)
    return _MappingLoss.apply(image, depth, marker, gt_image, gt_depth, kp, float(rgb_boundary_threshold), exposure)


def mapping_loss(config, image, depth, marker, viewpoint, initialization: bool = False):
    """get_loss_mapping(config, image, depth, viewpoint, opacity, initialization) +
    get_loss_marker(config, marker, viewpoint.kp_score), one pass."""
    dev = image.device
    gt_depth = viewpoint.depth
    if isinstance(gt_depth, np.ndarray):
        gt_depth = torch.from_numpy(gt_depth)
    gt_depth = gt_depth.to(dtype=torch.float32, device=dev)
    thr = config["Training"]["rgb_boundary_threshold"]
    a = None if initialization else viewpoint.exposure_a
    b = None if initialization else viewpoint.exposure_b
    return mapping_loss_tensors(image, depth, marker, viewpoint.original_image.to(dev), gt_depth,
                                viewpoint.kp_score.to(dev), thr, a, b)


def mapping_loss_window(config, pkgs, viewpoints, initialization: bool = False):
    """The per-view mapping losses of a window WITHOUT autograd nodes of their own: every view's fused launch already holds
    the gradients w.r.t. its rendered buffers (the loss terms are summed with weight 1, train_gaussians.py:217-218), so the
    caller hands them to ONE torch.autograd.backward on the rasterizer's outputs instead of building, per view, a loss node,
    three `g * grad` products and an addition (≈ 9 small launches per view in map_step).

    Returns (tensors, grads, value): `tensors` / `grads` for torch.autograd.backward(tensors, grads) — the rendered buffers of
    every view and their gradients — and `value`, the summed loss as a 0-dim tensor (no graph).  The gradients of the exposure
    parameters (when a view has them and they require grad) are accumulated into their `.grad` here."""
    if not pkgs:
        return [], [], None
    if len(pkgs) != len(viewpoints):
        raise RuntimeError("mapping_loss_window: one view per render package (filter the two lists together)")
    dev = pkgs[0]["render"].device
    thr = config["Training"]["rgb_boundary_threshold"]
    V = len(pkgs)
    table = torch.empty((V, 4), dtype=torch.float32, device=dev)
    ex_all = None
    if not initialization and all(getattr(vp, "exposure_a", None) is not None for vp in viewpoints):
        ex_all = torch.cat([t.reshape(1) for vp in viewpoints for t in (vp.exposure_a, vp.exposure_b)]).to(torch.float32).detach()
    tensors, grads, pending = [], [], []
    with torch.no_grad():
        lib = _native.load()
        H, W = int(pkgs[0]["render"].shape[-2]), int(pkgs[0]["render"].shape[-1])
        HW = H * W
        same = all(tuple(p["render"].shape) == (3, H, W) and p["depth"].numel() == HW and p["kp_prob"].numel() == HW for p in pkgs)
        if same and V <= _native.MAX_WINDOW_VIEWS:
            # the window's per-view launches as ONE launch pair (splatraster_mapping_loss_window; per-view results bit-identical):
            # the gradients of all views are slices of one allocation [V, 5, H, W] = (g_image[3] | g_depth | g_marker) per view
            gbuf = torch.empty((V, 5, H, W), dtype=torch.float32, device=dev)
            ws_one = int(lib.splatraster_mapping_loss_workspace_bytes(HW))
            ws = torch.empty((V * ws_one,), dtype=torch.uint8, device=dev)
            lv = (_native.LossView * V)()
            keep = []      # prepared inputs stay alive until the launch is enqueued
            for v, (pkg, vp) in enumerate(zip(pkgs, viewpoints)):
                gt_depth = vp.depth
                if isinstance(gt_depth, np.ndarray):
                    gt_depth = torch.from_numpy(gt_depth)
                a = None if initialization else getattr(vp, "exposure_a", None)
                ex = ex_all[2 * v:2 * v + 2] if ex_all is not None else (
                    None if a is None else torch.cat((vp.exposure_a.reshape(1), vp.exposure_b.reshape(1))).to(torch.float32).detach())
                ins = [_prep(t, dev) for t in (pkg["render"], pkg["depth"], pkg["kp_prob"], vp.original_image, gt_depth,
                                                 vp.kp_score.to(torch.float32))]
                if any(t is None for t in ins):
                    raise RuntimeError("mapping_loss_window: empty input")
                exp_t = None     # (two floats read with scalar loads: no alignment requirement, so a slice of the window's table is passed as it is)
                if ex is not None:
                    ok = ex.dtype is torch.float32 and ex.device == dev and ex.is_contiguous()
                    exp_t = ex.detach() if ok else ex.detach().to(device=dev, dtype=torch.float32).contiguous()
                keep.append((ins, exp_t))
                g_image, g_depth, g_marker = gbuf[v, 0:3], gbuf[v, 3:4].view(pkg["depth"].shape), gbuf[v, 4].view(pkg["kp_prob"].shape)
                e = lv[v]
                e.image, e.depth, e.marker, e.gt_image, e.gt_depth, e.kp = (t.data_ptr() for t in ins)
                e.exposure = exp_t.data_ptr() if exp_t is not None else None
                e.g_image, e.g_depth, e.g_marker = g_image.data_ptr(), g_depth.data_ptr(), g_marker.data_ptr()
                tensors += [pkg["render"], pkg["depth"], pkg["kp_prob"]]
                grads += [g_image, g_depth, g_marker]
                if ex is not None:
                    pending += [(prm, v, col) for prm, col in ((vp.exposure_a, 2), (vp.exposure_b, 3))
                                if isinstance(prm, torch.Tensor) and prm.requires_grad]
            with _on_device(dev):
                _native.check(lib.splatraster_mapping_loss_window(V, HW, lv, C.c_float(float(thr)), _ptr(table), _ptr(ws), _stream(dev)),
                              "mapping_loss_window")
            del keep
        else:
          for v, (pkg, vp) in enumerate(zip(pkgs, viewpoints)):
            gt_depth = vp.depth
            if isinstance(gt_depth, np.ndarray):
                gt_depth = torch.from_numpy(gt_depth)
            gt_depth = gt_depth.to(dtype=torch.float32, device=dev)
            a = None if initialization else getattr(vp, "exposure_a", None)
            ex = ex_all[2 * v:2 * v + 2] if ex_all is not None else (
                None if a is None else torch.cat((vp.exposure_a.reshape(1), vp.exposure_b.reshape(1))).to(torch.float32))
            g_image, g_depth, g_marker, out = _mapping_loss_launch(
                pkg["render"].detach(), pkg["depth"].detach(), pkg["kp_prob"].detach(), vp.original_image.to(dev), gt_depth,
                vp.kp_score.to(dev), thr, ex, out=table[v])
            tensors += [pkg["render"], pkg["depth"], pkg["kp_prob"]]
            grads += [g_image, g_depth, g_marker]
            if ex is not None:
                pending += [(prm, v, col) for prm, col in ((vp.exposure_a, 2), (vp.exposure_b, 3))
                            if isinstance(prm, torch.Tensor) and prm.requires_grad]
        if pending:
            # the exposure gradients leave the table in ONE copy (a 4-byte clone per parameter was two 5-us copy operations per view
            # in the stream of a map step); every parameter's .grad is its own element of that copy: an in-place op on one of them
            # (clip_grad_norm_, ...) touches neither the table nor another parameter's gradient
            gex = table[:, 2:4].clone()
            for prm, v, col in pending:
                gpart = gex[v, col - 2:col - 1].reshape(prm.shape)
                prm.grad = gpart if prm.grad is None else prm.grad + gpart
        value = table[:, :2].sum()
    return tensors, grads, value


def _refinement_loss_launch(image, gt, lam: float):
    """The fused L1 + SSIM launch: (g_image, out) with out = [L1, SSIM, lam-weighted loss]; g_image = d out[2] / d image."""
    lib = _native.load()
    _require_gpu(image, "image")
    dev = image.device
    if image.dim() != 3 or tuple(image.shape) != tuple(gt.shape):
        raise RuntimeError("refinement_loss: expected image and gt of the same [C,H,W] shape")
    Cn, H, W = (int(v) for v in image.shape)
    im, g = _prep(image, dev), _prep(gt, dev)
    g_image = torch.empty((Cn, H, W), dtype=torch.float32, device=dev)
    out = torch.empty((3,), dtype=torch.float32, device=dev)
    ws = torch.empty((lib.splatraster_refinement_loss_workspace_bytes(Cn, H, W),), dtype=torch.uint8, device=dev)
    with _on_device(dev):
        _native.check(lib.splatraster_refinement_loss(Cn, H, W, C.c_float(lam), _ptr(im), _ptr(g), _ptr(g_image),
                                                      _ptr(out), _ptr(ws), _stream(dev)), "refinement_loss")
    return g_image, out


def refinement_loss_and_grad(image, gt, lambda_dssim: float = 0.2):
    """(value, d value / d image) of the colour-refinement loss without an autograd node: the caller runs
    `image.backward(grad)` (color_refinement_step) — no `g * grad` product for an upstream gradient that is always 1."""
    with torch.no_grad():
        g_image, out = _refinement_loss_launch(image.detach(), gt, float(lambda_dssim))
    return out[2], g_image


class _RefinementLoss(torch.autograd.Function):
    """mode 0: (1 - lambda) L1 + lambda (1 - SSIM);  mode 1: SSIM;  mode 2: L1."""

    @staticmethod
    def forward(ctx, image, gt, lambda_dssim: float, mode: int):
        lam = {0: float(lambda_dssim), 1: 1.0, 2: 0.0}[mode]
        g_image, out = _refinement_loss_launch(image, gt, lam)
        ctx.save_for_backward(g_image)
        ctx.sign = -1.0 if mode == 1 else 1.0     # the kernel's gradient is that of lambda (1 - ssim)
        return out[{0: 2, 1: 1, 2: 0}[mode]]

    @staticmethod
    def backward(ctx, g):
        (g_image,) = ctx.saved_tensors
        return (ctx.sign * g) * g_image, None, None, None


def refinement_loss(image, gt, lambda_dssim: float = 0.2):
    """(1 - lambda_dssim) * l1_loss(image, gt) + lambda_dssim * (1 - ssim(image, gt)) — the
    colour-refinement loss of train_gaussians.py:283-285 — value and image gradient in two HIP
    kernels instead of five 11x11 grouped convolutions and their autograd backward."""
    return _RefinementLoss.apply(image, gt, float(lambda_dssim), 0)


def ssim(img1, img2, window_size: int = 11, size_average: bool = True):
    """gaussian_splatting/utils/loss_utils.py:61-70 (the configuration SplatLoc uses)."""
    if window_size != 11 or not size_average:
        raise RuntimeError("ssim: only window_size=11, size_average=True is implemented on the HIP path")
    return _RefinementLoss.apply(img1, img2, 1.0, 1)


def l1_loss(network_output, gt):
    """gaussian_splatting/utils/loss_utils.py:21-22."""
    return _RefinementLoss.apply(network_output, gt, 0.0, 2)


class _IsotropicLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scaling, marker):
        lib = _native.load()
        _require_gpu(scaling, "scaling")
        dev = scaling.device
        P, SC = int(scaling.shape[0]), int(scaling.shape[1])
        if SC not in (1, 3) or marker.numel() != P:
            raise RuntimeError("isotropic_loss: expected scaling [P,3] (or [P,1]) and marker [P,1]")
        s, mk = _prep(scaling, dev), _prep(marker, dev)
        row_grad = torch.empty((P,), dtype=torch.float32, device=dev)
        out = torch.empty((2,), dtype=torch.float32, device=dev)
        ws = torch.empty((lib.splatraster_isotropic_loss_workspace_bytes(P),), dtype=torch.uint8, device=dev)
        with _on_device(dev):
            _native.check(lib.splatraster_isotropic_loss(P, SC, _ptr(s), _ptr(mk), _ptr(row_grad), _ptr(out), _ptr(ws),
                                                         _stream(dev)), "isotropic_loss")
        ctx.save_for_backward(row_grad, out)
        ctx.SC = SC
        return out[0]

    @staticmethod
    def backward(ctx, g):
        row_grad, out = ctx.saved_tensors
        return ((g * out[1]) * row_grad).view(-1, 1).expand(-1, ctx.SC), None


def isotropic_loss(scaling, marker):
    """The isotropic scale regulariser of SplatLoc.map (train_gaussians.py:222-226):

        mask = marker.detach().squeeze() > 0.005
        torch.abs(scaling.mean(dim=1).view(-1, 1)[mask] / (0.02 * (1 - marker[mask])) - 1).mean()

    value and gradient w.r.t. the ACTIVATED `scaling` in two small HIP launches, with no `.cpu()`
    synchronisation for the mask.  An empty mask gives 0 (the reference's mean over nothing is NaN)."""
    return _IsotropicLoss.apply(scaling, marker.detach())
