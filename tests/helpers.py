"""Shared helpers of the parity tests: run a Scene through the oracle / the HIP path."""
from __future__ import annotations

import numpy as np
import torch

from oracle import oracle
from splatloc_amd.synthetic import Scene


def oracle_forward(sc: Scene, omp: bool = True, **over):
    cam = sc.camera
    st = oracle.Settings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy,
                         scale_modifier=over.pop("scale_modifier", 1.0), sh_degree=over.pop("sh_degree", 0))
    kw = dict(colors_precomp=sc.features.cpu().numpy(), scales=sc.scales.cpu().numpy(),
              rotations=sc.rotations.cpu().numpy())
    kw.update(over)
    return oracle.forward(st, sc.bg.cpu().numpy(), sc.means3D.cpu().numpy(), sc.opacities.cpu().numpy(),
                          cam.world_view_transform.cpu().numpy(), cam.full_proj_transform.cpu().numpy(),
                          cam.camera_center.cpu().numpy(), omp=omp, **kw)


def oracle_backward(fwd, sc: Scene, use_depth=True, use_alpha=True, omp: bool = True):
    return oracle.backward(fwd, sc.dL_dcolor.cpu().numpy(),
                           sc.dL_ddepth.cpu().numpy() if use_depth else None,
                           sc.dL_dalpha.cpu().numpy() if use_alpha else None, omp=omp)


def hip_settings(sc: Scene, device, scale_modifier=1.0, sh_degree=0):
    from splatloc_amd import GaussianRasterizationSettings
    cam = sc.camera
    return GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
        bg=sc.bg.to(device), scale_modifier=scale_modifier, viewmatrix=cam.world_view_transform.to(device),
        projmatrix=cam.full_proj_transform.to(device), sh_degree=sh_degree,
        campos=cam.camera_center.to(device), prefiltered=False, debug=False)


class HipRun:
    """One forward (+ optional backward) of the HIP path with every intermediate exposed."""

    def __init__(self, sc: Scene, device="cuda:0", scale_modifier=1.0, sh_degree=0, shs=None, cov3D=None,
                 backward=True, use_depth=True, use_alpha=True):
        from splatloc_amd import GaussianRasterizer
        from splatloc_amd import introspect
        dev = torch.device(device)
        leaf = lambda t: t.to(dev).clone().requires_grad_(True)  # noqa: E731
        self.means3D = leaf(sc.means3D)
        self.means2D = torch.zeros_like(self.means3D, requires_grad=True)
        self.opacities = leaf(sc.opacities)
        self.colors = leaf(sc.features) if shs is None else None
        self.shs = leaf(shs) if shs is not None else None
        self.scales = leaf(sc.scales) if cov3D is None else None
        self.rotations = leaf(sc.rotations) if cov3D is None else None
        self.cov3D = leaf(cov3D) if cov3D is not None else None
        rs = hip_settings(sc, dev, scale_modifier, sh_degree)
        rast = GaussianRasterizer(raster_settings=rs)
        color, depth, alpha, radii = rast(means3D=self.means3D, means2D=self.means2D, shs=self.shs,
                                          colors_precomp=self.colors, opacities=self.opacities,
                                          scales=self.scales, rotations=self.rotations,
                                          cov3D_precomp=self.cov3D)
        self.color, self.depth, self.alpha, self.radii = color, depth, alpha, radii
        fn = color.grad_fn
        saved = fn.saved_tensors
        geom, binning, img = saved[12], saved[13], saved[14]
        P = sc.means3D.shape[0]
        W, H = sc.camera.image_width, sc.camera.image_height
        self.num_rendered = fn.num_rendered
        self.state = introspect.forward_state((geom, binning, img), P, W, H, fn.num_rendered)
        if backward:
            loss = (color * sc.dL_dcolor.to(dev)).sum()
            if use_depth:
                loss = loss + (depth * sc.dL_ddepth.to(dev)).sum()
            if use_alpha:
                loss = loss + (alpha * sc.dL_dalpha.to(dev)).sum()
            loss.backward()
        torch.cuda.synchronize(dev)

    def np(self, t):
        return None if t is None else t.detach().cpu().numpy()


def assert_grad_close(name, got, ref, rtol=2e-3, atol_scale=1e-4, allow_frac=0.0, outlier_factor=30.0):
    """|got - ref| <= rtol*|ref| + atol_scale*max|ref| elementwise (atomics reorder sums).
    allow_frac > 0 (full-size frames only): that fraction of the elements may miss the bar — the
    Gaussians behind an alpha >= 1/255 test that flipped within an ulp — but by no more than
    outlier_factor times the tolerance."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, f"{name}: shape {got.shape} vs {ref.shape}"
    scale = np.abs(ref).max() if ref.size else 0.0
    tol = rtol * np.abs(ref) + atol_scale * scale + 1e-30
    bad = np.abs(got - ref) > tol
    if allow_frac > 0.0 and bad.any():
        assert bad.mean() <= allow_frac and not (np.abs(got - ref) > outlier_factor * tol).any(), (
            f"{name}: {bad.sum()} / {bad.size} elements off (allowed fraction {allow_frac}); worst abs err "
            f"{np.abs(got - ref).max():.3e} (scale {scale:.3e})")
        return
    assert not bad.any(), (f"{name}: {bad.sum()} / {bad.size} elements off; worst abs err "
                           f"{np.abs(got - ref).max():.3e} (scale {scale:.3e})")


def assert_grad_rows_close(name, got, ref, rtol=1e-4, row_atol=1e-3, allow_frac=0.0, outlier_factor=10.0):
    """Per-ROW bar: |got - ref| <= rtol * |ref| + row_atol * max|ref[row]| elementwise — the absolute term is relative to the
    Gaussian's OWN gradient row (a [P, k] tensor has P rows; a [P, 1] tensor makes it a purely relative bar), so a row a
    thousand times smaller than the tensor's maximum cannot hide behind it.  Rows whose reference is exactly zero must be
    exactly zero.  `allow_frac`: fraction of the ELEMENTS that may miss the bar, by at most `outlier_factor` times the
    tolerance (cancellation: a row sum that is itself the small difference of large pixel terms)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, f"{name}: shape {got.shape} vs {ref.shape}"
    g2, r2 = got.reshape(got.shape[0], -1), ref.reshape(ref.shape[0], -1)
    rowmax = np.abs(r2).max(axis=1, keepdims=True) if r2.size else np.zeros((0, 1))
    tol = rtol * np.abs(r2) + row_atol * rowmax
    err = np.abs(g2 - r2)
    bad = err > tol
    if not bad.any():
        return
    frac = bad.mean()
    worst = (err / np.maximum(tol, 1e-300))[bad].max() if (tol[bad] > 0).all() else np.inf
    if not np.isfinite(outlier_factor):
        worst = 0.0       # counted, not bounded (one-element rows: a relative bar on a cancelling sum)
    assert frac <= allow_frac and worst <= outlier_factor, (
        f"{name}: {bad.sum()} / {bad.size} elements ({frac:.2e}) beyond rtol {rtol} + {row_atol} x row max "
        f"(allowed fraction {allow_frac}); worst {worst:.1f} x the tolerance")
