"""simple_knn._C.distCUDA2 replacement (gaussian_splatting/scene/gaussian_model.py:18,206)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _native
from .rasterizer import _on_device


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """Mean squared distance of every point to its 3 nearest other points. [N,3] -> [N]."""
    lib = _native.load()
    if not points.is_cuda:
        raise RuntimeError("distCUDA2: points must be on a ROCm device (no CPU fallback)")
    dev = points.device
    pts = points.detach().to(dtype=torch.float32).contiguous()
    N = int(pts.shape[0])
    out = torch.zeros((N,), dtype=torch.float32, device=dev)
    if N == 0:
        return out
    ws = torch.empty((lib.splatknn_workspace_bytes(N),), dtype=torch.uint8, device=dev)
    with _on_device(dev):
        _native.check(lib.splatknn_dist2(N, C.c_void_p(pts.data_ptr()), C.c_void_p(out.data_ptr()),
                                         C.c_void_p(ws.data_ptr()),
                                         C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "distCUDA2")
    return out
