"""`eval_rendering` on the device (utils/eval_utils.py:22-72 of the reference; BASELINE config 4's stand-in).

The reference renders every test frame under `no_grad`, clamps the image to [0, 1] and scores it against the ground
truth: PSNR over the elements where `gt > 0` (gaussian_splatting/utils/image_utils.py:19-21), SSIM over the whole frame
(loss_utils.py:61-102) and LPIPS (a learned AlexNet metric from torchmetrics — a network, not part of this library:
`mean_lpips` is reported as None).  Here the frames go through the rasterizer as forward-only WINDOWS (one launch
sequence per `window` frames, splatloc_amd.fused.render_window) and every frame's clamp + mask + squared error + SSIM
map is ONE kernel + a one-block finish (`splatraster_eval_metrics`); nothing is copied to the host until the caller
reads the result tensors.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _native
from .fused import render_window
from .rasterizer import _on_device, _prep, _ptr, _require_gpu, _stream


def eval_metrics(render: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """One frame: out[4] = (psnr, ssim, masked mse, mask count) of clamp(render, 0, 1) against gt — a device tensor,
    no host synchronisation.  render, gt: [C,H,W] float32 on the ROCm device."""
    lib = _native.load()
    _require_gpu(render, "render")
    dev = render.device
    if render.dim() != 3 or tuple(render.shape) != tuple(gt.shape):
        raise RuntimeError("eval_metrics: expected render and gt of the same [C,H,W] shape")
    Cn, H, W = (int(v) for v in render.shape)
    im, g = _prep(render, dev), _prep(gt, dev)
    out = torch.empty((4,), dtype=torch.float32, device=dev)
    ws = torch.empty((lib.splatraster_eval_metrics_workspace_bytes(Cn, H, W),), dtype=torch.uint8, device=dev)
    with _on_device(dev):
        _native.check(lib.splatraster_eval_metrics(Cn, H, W, _ptr(im), _ptr(g), _ptr(out), _ptr(ws), _stream(dev)),
                      "eval_metrics")
    return out


def eval_rendering(frames, gaussians, gt_images, pipe, background, window: int = 5) -> dict:
    """utils/eval_utils.py:22-72 without the dataset object: `frames` are the cameras, `gt_images[k]` the ground-truth
    [3,H,W] image of frame k (None = the reference's `valid == False`: skipped).  Returns
    {"mean_psnr", "mean_ssim", "mean_lpips": None, "psnr": [...], "ssim": [...], "frames": n} — ONE device->host read
    at the end (the reference: three `.item()` per frame)."""
    frames = list(frames)
    pairs = [(f, g) for f, g in zip(frames, gt_images) if g is not None]
    rows = []
    with torch.no_grad():
        for a in range(0, len(pairs), max(int(window), 1)):
            chunk = pairs[a:a + max(int(window), 1)]
            pkgs, _ = render_window([f for f, _ in chunk], gaussians, pipe, background)
            for pkg, (_, gt) in zip(pkgs, chunk):
                if pkg is None:
                    continue
                rows.append(eval_metrics(pkg["render"], gt.to(pkg["render"].device)))
        table = torch.stack(rows).cpu() if rows else torch.zeros((0, 4))
    psnr, ssim = table[:, 0].tolist(), table[:, 1].tolist()
    n = len(psnr)
    return {"mean_psnr": (sum(psnr) / n) if n else float("nan"), "mean_ssim": (sum(ssim) / n) if n else float("nan"),
            "mean_lpips": None, "psnr": psnr, "ssim": ssim, "frames": n}
