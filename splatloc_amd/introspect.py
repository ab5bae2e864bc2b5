"""Typed views into the opaque forward-state buffers (debug / parity tests).

Uses the layout queries of include/splatraster.h, so tests can compare the integer
intermediates (tile counts, depth order, sorted point list, tile ranges, n_contrib)
bit-for-bit with the oracle.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _native


def _view(buf: torch.Tensor, off: int, count: int, dtype: torch.dtype) -> torch.Tensor:
    nbytes = count * torch.empty((), dtype=dtype).element_size()
    return buf[off:off + nbytes].view(dtype)


def geometry_views(geom: torch.Tensor, P: int) -> dict:
    lib = _native.load()
    L = _native.GeometryLayout()
    _native.check(lib.splatraster_get_geometry_layout(P, C.byref(L)), "geometry_layout")
    return dict(
        rec0=_view(geom, L.rec0, 8 * P, torch.float32).view(P, 8)[:, :4],
        rec1=_view(geom, L.rec0, 8 * P, torch.float32).view(P, 8)[:, 4:],
        tiles_touched=_view(geom, L.tiles_touched, P, torch.int32),
        depth_order=_view(geom, L.depth_order, P, torch.int32) & 0xFFFFFF,   # (bits 24..31: min(tiles_touched, 255))
        offsets=_view(geom, L.offsets, P, torch.int32),
        rgb=_view(geom, L.rgb, 3 * P, torch.float32).view(P, 3),
        clamped=_view(geom, L.clamped, 3 * P, torch.uint8).view(P, 3),
    )


def binning_views(binning: torch.Tensor, P: int, R: int, W: int, H: int, channels: int = 4) -> dict:
    lib = _native.load()
    L = _native.BinningLayout()
    _native.check(lib.splatraster_get_binning_layout(P, R, W, H, channels, C.byref(L)), "binning_layout")
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    return dict(
        point_list=_view(binning, L.point_list, R, torch.int32),
        tile_list=_view(binning, L.tile_list, R, torch.int32),
        ranges=_view(binning, L.ranges, 2 * tiles, torch.int32).view(tiles, 2),
    )


def image_views(img: torch.Tensor, W: int, H: int) -> dict:
    lib = _native.load()
    L = _native.ImageLayout()
    _native.check(lib.splatraster_get_image_layout(W, H, C.byref(L)), "image_layout")
    return dict(
        final_T=_view(img, L.final_T, W * H, torch.float32).view(H, W),
        n_contrib=_view(img, L.n_contrib, W * H, torch.int32).view(H, W),
    )


def forward_state(fn_ctx_tensors, P: int, W: int, H: int, R: int) -> dict:
    """Views for the (geom, binning, img) tensors saved by _RasterizeGaussians.forward."""
    geom, binning, img = fn_ctx_tensors
    out = {}
    out.update(geometry_views(geom, P))
    out.update(binning_views(binning, P, R, W, H))
    out.update(image_views(img, W, H))
    return out


def window_state(fn_ctx_tensors, P: int, V: int, W: int, H: int, R_per_view) -> list:
    """Per-view views into the (geom, binning, img) buffers of a _RasterizeWindow forward, translated to what the
    per-view call exposes: rows -> Gaussian indices, global tiles -> the view's tiles, ranges relative to the view's
    own (contiguous) segment of the instance list."""
    lib = _native.load()
    geom, binning, img = fn_ctx_tensors
    Rt = int(sum(R_per_view))
    GL, BL, IL = _native.GeometryLayout(), _native.BinningLayout(), _native.ImageLayout()
    _native.check(lib.splatraster_get_window_geometry_layout(P, V, C.byref(GL)), "window_geometry_layout")
    _native.check(lib.splatraster_get_window_binning_layout(P, V, Rt, W, H, 4, C.byref(BL)), "window_binning_layout")
    _native.check(lib.splatraster_get_window_image_layout(W, H, V, C.byref(IL)), "window_image_layout")
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    n = P * V
    rec = _view(geom, GL.rec0, 8 * n, torch.float32).view(V, P, 8)
    tt = _view(geom, GL.tiles_touched, n, torch.int32).view(V, P)
    pl = _view(binning, BL.point_list, Rt, torch.int32)
    tl = _view(binning, BL.tile_list, Rt, torch.int32)
    rng = _view(binning, BL.ranges, 2 * V * tiles, torch.int32).view(V, tiles, 2)
    fT = _view(img, IL.final_T, V * W * H, torch.float32).view(V, H, W)
    nc = _view(img, IL.n_contrib, V * W * H, torch.int32).view(V, H, W)
    out, start = [], 0
    for v in range(V):
        Rv = int(R_per_view[v])
        r = rng[v].clone()
        nz = r[:, 1] > r[:, 0]
        r[nz] -= start
        out.append(dict(rec0=rec[v, :, :4], rec1=rec[v, :, 4:], tiles_touched=tt[v], point_list=pl[start:start + Rv] - v * P,
                        tile_list=tl[start:start + Rv] - v * tiles, ranges=r, final_T=fT[v], n_contrib=nc[v]))
        start += Rv
    return out
