"""Fused front-end of the rasterizer call: SplatLoc's `render()` with the parameter
activations and the SH / feature packing done by ONE HIP kernel each way.

`render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color, mask)` has the
signature, semantics and return dict of the reference's
gaussian_splatting/gaussian_renderer/__init__.py:13-141; it reads the raw optimiser tensors of
the GaussianModel (`_xyz, _features_dc, _features_rest, _scaling, _rotation, _opacity,
_kp_score`, gaussian_model.py:40-55) and produces the rasterizer arguments with
`activate_pack` instead of the reference's chain of elementwise torch ops
(gaussian_model.py:78-105, gaussian_renderer/__init__.py:73-102, sh_utils.py:55-118):

    from splatloc_amd.fused import render          # instead of gaussian_renderer.render

Gradients reach the same leaves with the same values (tests/test_gpu_activations.py, against
the fixture recorded from the reference's own autograd).  Not supported by the fused path
(raises instead of silently diverging): SH degree 4, a camera centre that requires grad.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch

from . import _native
from .rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, _on_device, _prep, _ptr, _require_gpu, _stream,
                         _window_compatible, rasterize_window)


class _ActivatePack(torch.autograd.Function):
    """(xyz, f_dc, f_rest, scaling, rotation, opacity, extra, campos) ->
    (scales [P,3], rotations [P,4], opacities [P,1], colors [P,3+E])."""

    @staticmethod
    def forward(ctx, xyz, f_dc, f_rest, scaling, rotation, opacity, extra, campos, active_sh_degree: int):
        lib = _native.load()
        _require_gpu(xyz, "xyz")
        dev = xyz.device
        P = int(xyz.shape[0])
        K = 1 + (0 if f_rest is None else int(f_rest.shape[1]))
        E = 0 if extra is None else int(extra.shape[1])
        SC = int(scaling.shape[1])
        if campos is not None and campos.requires_grad:
            raise RuntimeError("activate_pack: the camera centre gets no gradient on the fused path")
        x, fd, fr, sc, ro, op, ex, cp = (_prep(t, dev) for t in (xyz, f_dc, f_rest, scaling, rotation, opacity,
                                                                 extra, campos))
        f32 = dict(dtype=torch.float32, device=dev)
        scales = torch.empty((P, 3), **f32)
        rotations = torch.empty((P, 4), **f32)
        opacities = torch.empty((P, 1), **f32)
        colors = torch.empty((P, 3 + E), **f32)
        with _on_device(dev):
            _native.check(lib.splatraster_activate_forward(
                P, K, int(active_sh_degree), SC, E, _ptr(x), _ptr(fd), _ptr(fr), _ptr(sc), _ptr(ro), _ptr(op),
                _ptr(ex), _ptr(cp), _ptr(scales), _ptr(rotations), _ptr(opacities), _ptr(colors), _stream(dev)),
                "activate_forward")
        ctx.cfg = (P, K, int(active_sh_degree), SC, E)
        ctx.shapes = (tuple(f_dc.shape), None if f_rest is None else tuple(f_rest.shape))
        ctx.save_for_backward(*[t if t is not None else torch.empty(0, device=dev) for t in
                                (x, fd, fr, sc, ro, op, cp)])
        return scales, rotations, opacities, colors

    @staticmethod
    def backward(ctx, g_scales, g_rotations, g_opacities, g_colors):
        lib = _native.load()
        x, fd, fr, sc, ro, op, cp = ctx.saved_tensors
        dev = x.device
        P, K, deg, SC, E = ctx.cfg
        opt = lambda t: t if t.numel() else None  # noqa: E731
        fr, cp = opt(fr), opt(cp)
        f32 = dict(dtype=torch.float32, device=dev)
        zeros = lambda shp: torch.zeros(shp, **f32)  # noqa: E731
        gs = _prep(g_scales, dev) if g_scales is not None else zeros((P, 3))
        gr = _prep(g_rotations, dev) if g_rotations is not None else zeros((P, 4))
        go = _prep(g_opacities, dev) if g_opacities is not None else zeros((P, 1))
        gc = _prep(g_colors, dev) if g_colors is not None else zeros((P, 3 + E))
        if P == 0:
            gs = gr = go = gc = None
        d_xyz = torch.empty((P, 3), **f32) if deg > 0 else None
        d_fd = torch.empty(ctx.shapes[0], **f32)
        d_fr = torch.empty(ctx.shapes[1], **f32) if ctx.shapes[1] is not None else None
        d_sc = torch.empty((P, SC), **f32)
        d_ro = torch.empty((P, 4), **f32)
        d_op = torch.empty((P, 1), **f32)
        d_ex = torch.empty((P, E), **f32) if E else None
        with _on_device(dev):
            _native.check(lib.splatraster_activate_backward(
                P, K, deg, SC, E, _ptr(x), _ptr(fd), _ptr(fr), _ptr(sc), _ptr(ro), _ptr(op), _ptr(cp),
                _ptr(gs), _ptr(gr), _ptr(go), _ptr(gc), _ptr(d_xyz), _ptr(d_fd),
                _ptr(d_fr) if (d_fr is not None and d_fr.numel()) else None, _ptr(d_sc), _ptr(d_ro), _ptr(d_op),
                _ptr(d_ex), _stream(dev)), "activate_backward")
        return d_xyz, d_fd, d_fr, d_sc, d_ro, d_op, d_ex, None, None


def activate_pack(xyz, f_dc, f_rest, scaling, rotation, opacity, extra=None, campos=None, active_sh_degree: int = 0):
    """Rasterizer arguments from raw parameters, one kernel (see module docstring).

    f_dc [P,1,3]; f_rest [P,K-1,3] or None; scaling [P,3] or [P,1]; extra [P,E] or None.
    Returns (scales, rotations, opacities, colors[P,3+E])."""
    if active_sh_degree < 0 or active_sh_degree > 3:
        raise RuntimeError("activate_pack: SH degree must be 0..3")
    # an empty f_rest ([P,0,3], SplatLoc's SH degree 0) stays an autograd input: it receives an empty gradient,
    # as it does from the reference's `torch.cat((features_dc, features_rest), dim=1)` (gaussian_model.py:96-100),
    # so torch.optim.Adam creates (empty) state for the group exactly as in the reference
    f_rest_arg = f_rest
    K = 1 + (0 if f_rest_arg is None else int(f_rest_arg.shape[1]))
    if (active_sh_degree + 1) ** 2 > K:
        raise RuntimeError("activate_pack: not enough SH coefficients for the active degree")
    if active_sh_degree > 0 and campos is None:
        raise RuntimeError("activate_pack: view-dependent colour needs the camera centre")
    out = _ActivatePack.apply(xyz, f_dc, f_rest_arg, scaling, rotation, opacity, extra, campos, int(active_sh_degree))
    return out


def _covariance_python(scales, scaling_modifier, rotation_raw):
    """GaussianModel.get_covariance (gaussian_model.py:72-76,110-111; general_utils build_rotation:
    the raw quaternion is normalised there): strip_symmetric(L L^T), L = R(q) diag(mod * s).
    Unfused torch ops on the device — only taken with pipe.compute_cov3D_python."""
    q = rotation_raw / rotation_raw.norm(dim=1, keepdim=True)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).view(-1, 3, 3)
    L = R * (scaling_modifier * scales).unsqueeze(1)
    cov = L @ L.transpose(1, 2)
    return torch.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], dim=1)


def shared_activations(pc, pipe):
    """The rasterizer arguments that do not depend on the view — `exp(_scaling)`, `normalize(_rotation)`, `sigmoid(_opacity)`
    and, at SH degree 0 (SplatLoc's configuration), the `[rgb | kp_score]` table — computed ONCE for all the views of a
    window (`render_window`): one `activate_pack` forward and, through autograd, one backward per optimisation step instead
    of one per view.  None when the colours are view dependent or the pipe asks for another path."""
    if not bool(pipe.convert_SHs_python) or bool(pipe.compute_cov3D_python) or int(pc.active_sh_degree) != 0:
        return None
    return activate_pack(pc._xyz, pc._features_dc, pc._features_rest, pc._scaling, pc._rotation, pc._opacity,
                         extra=pc._kp_score, campos=None, active_sh_degree=0)


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None,
           mask=None, _shared=None) -> Optional[dict]:
    """Drop-in for gaussian_renderer.render (gaussian_renderer/__init__.py:13-141): same signature, same
    return dict, same values and gradients (tests/test_gpu_activations.py against the reference's own render()).

    Two corners where the reference's code path is dead or broken, and what happens here:
      * `override_color` is IGNORED, exactly as in the reference (`colors_precomp` is reset to None right before
        the `if colors_precomp is None` test, gaussian_renderer/__init__.py:83-98, so the `else` is unreachable);
      * `mask` with `pipe.convert_SHs_python=True` raises TypeError in the reference (`shs[mask]` with shs = None,
        :108); here it renders the masked subset with the [rgb | kp_score] table like the unmasked call.  With
        `convert_SHs_python=False` both render 3 SH channels and `kp_prob` is channel 2, as in the reference."""
    xyz = pc._xyz
    if xyz.shape[0] == 0:
        return None
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=xyz.device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:  # noqa: BLE001
        pass
    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    campos = viewpoint_camera.camera_center
    rs = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree, campos=campos, prefiltered=False, debug=False)
    rasterizer = GaussianRasterizer(raster_settings=rs)

    sel = (lambda t: t) if mask is None else (lambda t: t[mask])
    means3D, means2D = sel(xyz), sel(screenspace_points)
    f_dc, f_rest = sel(pc._features_dc), sel(pc._features_rest)
    convert_shs = bool(pipe.convert_SHs_python)
    # the packed colours are only needed when the rasterizer is fed colors_precomp
    want_colors = convert_shs
    if _shared is not None and mask is None:
        scales, rotations, opacity, colors = _shared      # shared_activations(): the same tensors for every view of a window
    else:
        scales, rotations, opacity, colors = activate_pack(
            means3D, f_dc, f_rest, sel(pc._scaling), sel(pc._rotation), sel(pc._opacity),
            extra=sel(pc._kp_score) if want_colors else None, campos=campos,
            active_sh_degree=pc.active_sh_degree if convert_shs else 0)
    shs = colors_precomp = cov3D_precomp = None
    if convert_shs:
        colors_precomp = colors
    else:
        shs = torch.cat((f_dc, f_rest), dim=1)
    if pipe.compute_cov3D_python:
        cov3D_precomp = _covariance_python(scales, scaling_modifier, sel(pc._rotation))
        scales = rotations = None
    rendered_image, depth, alpha, radii = rasterizer(
        means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp, opacities=opacity,
        scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    return {"render": rendered_image[:3, :, :], "kp_prob": rendered_image[-1, :, :],
            "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii,
            "depth": depth, "opacity": alpha}


_WINDOW_STREAMS: dict = {}


def window_streams(device, n: int):
    """`n` HIP streams (cached per device) for render_window."""
    key = (torch.device(device).index, n)
    if key not in _WINDOW_STREAMS:
        _WINDOW_STREAMS[key] = [torch.cuda.Stream(device=device) for _ in range(n)]
    return _WINDOW_STREAMS[key]


def _view_settings(vp, pc, bg_color, scaling_modifier) -> GaussianRasterizationSettings:
    return GaussianRasterizationSettings(
        image_height=int(vp.image_height), image_width=int(vp.image_width), tanfovx=math.tan(vp.FoVx * 0.5),
        tanfovy=math.tan(vp.FoVy * 0.5), bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=vp.world_view_transform, projmatrix=vp.full_proj_transform, sh_degree=pc.active_sh_degree,
        campos=vp.camera_center, prefiltered=False, debug=False)


def render_window(viewpoints, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, streams: int = 1, per_view=None,
                  share_activations: bool = True, batched: bool = True):
    """The render loop of one optimisation window (`for cam_idx in ...: render_pkg = render(viewpoint, ...)`,
    train_gaussians.py:195-219) with view k enqueued on HIP stream k % `streams`.

    The views of a window are independent until their losses are summed, and at SplatLoc's own frame size
    (640x480, C = 4) one frame's kernels leave most of the MI355X idle (a 1200-tile frame is 4800 waves for
    8192 wave slots; the kernel lasts as long as its longest list).  Spreading the views over a few streams
    lets another view's kernels fill the machine: +30 % frames/s at 640x480 with 2 streams, +7 % at 1080p
    (DESIGN.md).  `per_view(k, viewpoint, pkg)` — e.g. the view's loss — runs on the view's stream as
    well; its return values are collected.  Autograd replays every backward on its forward's stream and
    serialises the accumulation into the shared parameters' .grad, so `loss.backward()` needs no change.
    `share_activations`: at SH degree 0 the activated scales / rotations / opacities and the `[rgb | kp_score]` table are
    the same for every view, so they are produced by ONE `activate_pack` per window (and differentiated once: autograd sums
    the views' gradients at its outputs) instead of one per view.
    `batched` (default): when the rasterizer arguments are the same for every view (SH degree 0 — SplatLoc's
    configuration — so `shared_activations` applies) and the views share one image size, the whole window goes through
    `rasterize_window`: ONE launch sequence for the V views (one preprocess, one depth sort, one tile sort keyed by
    (view, tile), one compositing grid) and ONE backward that sums the views' parameter gradients in-kernel.  Per-view
    results are bit-identical to the loop; `streams` is ignored on this path (there is nothing left to overlap).
    Returns (pkgs, per_view results).  batched=False, streams <= 1 is the reference's serial loop."""
    viewpoints = list(viewpoints)
    pkgs, extra = [], []
    if not viewpoints:      # a rank without work in a frame-parallel step (world size > window size): nothing to launch
        return pkgs, extra
    dev = pc._xyz.device
    # view-independent activations once per window (on the caller's stream, before the fork)
    shared = shared_activations(pc, pipe) if (share_activations and (len(viewpoints) > 1 or batched) and pc._xyz.shape[0] > 0) else None
    if batched and shared is not None:
        settings = [_view_settings(vp, pc, bg_color, scaling_modifier) for vp in viewpoints]
        if _window_compatible(settings):
            scales, rotations, opacity, colors = shared
            xyz = pc._xyz
            # render(): screenspace_points, the per-view gradient carrier (zeros whose .grad the densification statistics
            # read).  One zero fill for the window instead of the reference's `zeros_like(...) + 0` pair of kernels per view:
            # each carrier is a leaf view of the block.
            block = torch.zeros((len(viewpoints),) + tuple(xyz.shape), dtype=xyz.dtype, device=xyz.device)
            carriers = [block[k].requires_grad_(True) for k in range(len(viewpoints))]
            # render = image[:3] and kp_prob = image[-1] leave the rasterizer as separate autograd outputs: their
            # gradients reach the backward kernel as separate planes, and an output the loss never touches costs nothing
            # (SplatLoc's layout [rgb | kp_score], C = 4; wider tables [rgb | features | kp_score]: the same two outputs, the
            #  feature channels stay inside the rasterizer — slicing one 35-channel output would cost two zero-filled
            #  [C,H,W] gradients and an addition per view in autograd)
            split = 3 if int(colors.shape[1]) >= 4 else False
            outs = rasterize_window(settings, xyz, carriers, colors, opacity, scales=scales, rotations=rotations,
                                    split_last=split)
            # visibility_filter = radii > 0 of every view in ONE launch when the views' radii are rows of one table
            rbase = outs[0][-1]._base if len(outs) > 1 else None
            vis_all = None
            if rbase is not None and rbase.dim() == 2 and rbase.shape[0] == len(outs) and all(
                    o[-1]._base is rbase and o[-1].data_ptr() == rbase[k].data_ptr() for k, o in enumerate(outs)):
                vis_all = rbase > 0
            for k, (vp, o) in enumerate(zip(viewpoints, outs)):
                if split:
                    rgb, kp_prob, depth, alpha, radii = o
                else:
                    img, depth, alpha, radii = o
                    rgb, kp_prob = img[:3, :, :], img[-1, :, :]
                pkg = {"render": rgb, "kp_prob": kp_prob, "viewspace_points": carriers[k],
                       "visibility_filter": vis_all[k] if vis_all is not None else radii > 0, "radii": radii, "depth": depth,
                       "opacity": alpha}
                pkgs.append(pkg)
                extra.append(per_view(k, vp, pkg) if per_view is not None else None)
            return pkgs, extra
    if streams <= 1 or len(viewpoints) <= 1:
        for k, vp in enumerate(viewpoints):
            pkg = render(vp, pc, pipe, bg_color, scaling_modifier, _shared=shared)
            pkgs.append(pkg)
            extra.append(per_view(k, vp, pkg) if per_view is not None else None)
        return pkgs, extra
    main = torch.cuda.current_stream(dev)
    side = window_streams(dev, min(streams, len(viewpoints)))
    for st in side:
        st.wait_stream(main)          # the parameters (and whatever else main produced) are ready
    for k, vp in enumerate(viewpoints):
        with torch.cuda.stream(side[k % len(side)]):
            pkg = render(vp, pc, pipe, bg_color, scaling_modifier, _shared=shared)
            pkgs.append(pkg)
            extra.append(per_view(k, vp, pkg) if per_view is not None else None)
    for st in side:
        main.wait_stream(st)          # join: everything above is visible to the caller's stream
    return pkgs, extra
