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
from .rasterizer import _prep, _ptr, _require_gpu, _stream


class _MappingLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, depth, marker, gt_image, gt_depth, kp, threshold: float, exposure):
        lib = _native.load()
        _require_gpu(image, "image")
        dev = image.device
        H, W = int(image.shape[-2]), int(image.shape[-1])
        HW = H * W
        if tuple(image.shape) != (3, H, W) or depth.numel() != HW or marker.numel() != HW:
            raise RuntimeError("mapping_loss: expected image [3,H,W], depth [1,H,W], marker [H,W]")
        im, de, ma, gi, gd = (_prep(t, dev) for t in (image, depth, marker, gt_image, gt_depth))
        k8 = kp.to(device=dev).ne(0).to(torch.uint8).contiguous() if kp.dtype != torch.uint8 else kp.to(dev).contiguous()
        ex = _prep(exposure, dev) if exposure is not None else None
        f32 = dict(dtype=torch.float32, device=dev)
        g_image = torch.empty((3, H, W), **f32)
        g_depth = torch.empty(tuple(depth.shape), **f32)
        g_marker = torch.empty(tuple(marker.shape), **f32)
        out = torch.empty((4,), **f32)
        ws = torch.empty((lib.splatraster_mapping_loss_workspace_bytes(HW),), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _native.check(lib.splatraster_mapping_loss(
                HW, _ptr(im), _ptr(de), _ptr(ma), _ptr(gi), _ptr(gd), _ptr(k8), C.c_float(float(threshold)), _ptr(ex),
                _ptr(g_image), _ptr(g_depth), _ptr(g_marker), _ptr(out), _ptr(ws), _stream(dev)), "mapping_loss")
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
    gt_depth [H,W], kp [H,W] bool; exposure_a / exposure_b one-element tensors or None (no affine)."""
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
