"""Device-side pieces of SplatLoc's densification bookkeeping."""
from __future__ import annotations

import torch

from . import _native
from .rasterizer import _ptr, _require_gpu, _stream


def add_densification_stats(viewspace_grad: torch.Tensor, radii: torch.Tensor, xyz_gradient_accum: torch.Tensor,
                            denom: torch.Tensor, max_radii2D: torch.Tensor) -> None:
    """In place, one launch, no host synchronisation — replaces, for one view,

        vis = radii > 0
        gaussians.max_radii2D[vis] = torch.max(gaussians.max_radii2D[vis], radii[vis])   # train_gaussians.py:240-244
        gaussians.add_densification_stats(viewspace_points, vis)                         # gaussian_model.py:677-679

    viewspace_grad = viewspace_points.grad [P,3]; radii int32 [P]; the three state tensors are
    float32 and contiguous (they are updated in place)."""
    _require_gpu(viewspace_grad, "viewspace_grad")
    dev = viewspace_grad.device
    P = int(viewspace_grad.shape[0])
    for t, name, n in ((xyz_gradient_accum, "xyz_gradient_accum", P), (denom, "denom", P), (max_radii2D, "max_radii2D", P)):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n or t.device != dev:
            raise RuntimeError(f"add_densification_stats: `{name}` must be a contiguous float32 tensor of {n} elements on {dev}")
    if radii.dtype != torch.int32 or radii.numel() != P or radii.device != dev:
        raise RuntimeError("add_densification_stats: `radii` must be the int32 [P] tensor the rasterizer returned")
    g = viewspace_grad.detach()
    if g.dtype != torch.float32 or not g.is_contiguous() or tuple(g.shape) != (P, 3):
        g = g.to(torch.float32).reshape(P, 3).contiguous()
    with torch.cuda.device(dev):
        _native.check(_native.load().splatraster_densification_stats(
            P, _ptr(g), _ptr(radii.contiguous()), _ptr(xyz_gradient_accum), _ptr(denom), _ptr(max_radii2D),
            _stream(dev)), "densification_stats")
