"""Host-side mirror of the `diff_gauss` extension API.

Names, argument meaning and error behaviour follow what SplatLoc calls at
gaussian_splatting/gaussian_renderer/__init__.py:42-57,117-126 (the extension's own
Python wrapper is un-vendored): `GaussianRasterizationSettings`
(12-field NamedTuple), `GaussianRasterizer(nn.Module)` whose forward returns the 4-tuple
`(color[C,H,W], depth[1,H,W], alpha[1,H,W], radii[P] int32)`, `rasterize_gaussians` and the
`_RasterizeGaussians` autograd.Function.

All compute is in the HIP library behind include/splatraster.h; tensors must live on a
ROCm device.  No CPU fallback: CPU tensors raise.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional

import torch
from torch import nn

from . import _native


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _prep(t: Optional[torch.Tensor], device) -> Optional[torch.Tensor]:
    """contiguous fp32 on `device`, 16-byte aligned (kernels use 128-bit loads).  The common case — the
    tensor already is all that — costs three attribute checks (host time is what bounds small frames)."""
    if t is None or t.numel() == 0:
        return None
    if t.dtype is torch.float32 and t.device == device and t.is_contiguous() and not (t.data_ptr() & 15):
        return t.detach() if t.requires_grad else t
    t = t.detach()
    if t.dtype != torch.float32 or t.device != device or not t.is_contiguous():
        t = t.to(device=device, dtype=torch.float32).contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


_EMPTY: dict = {}


def _empty(device) -> torch.Tensor:
    """one shared zero-element placeholder per device for the `None` slots of save_for_backward"""
    e = _EMPTY.get(device)
    if e is None:
        e = _EMPTY[device] = torch.empty(0, device=device)
    return e


class _on_device:
    """`with torch.cuda.device(dev)` only when `dev` is not already current (the context manager costs ~10 us)."""

    def __init__(self, device):
        self.ctx = None if torch.cuda.current_device() == device.index else torch.cuda.device(device)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device) -> C.c_void_p:
    """the hipStream_t torch would launch on right now (device's current stream).  The raw getter costs ~0.3 us; building a
    torch.cuda.Stream object ~8 us — five of those per refinement iteration were 10 % of its host time."""
    if _RAW_STREAM is not None:
        idx = device.index
        return C.c_void_p(_RAW_STREAM(torch.cuda.current_device() if idx is None else idx))
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class PlainCtx:
    """Stands in for the autograd context when `_RasterizeWindow.forward / .backward` (or `_ActivatePack`'s) are called
    DIRECTLY, outside autograd (training.color_refinement_step's direct path): the same code, the same kernels, no graph
    nodes, no engine thread hand-off."""
    needs_input_grad = (True,) * 16

    def save_for_backward(self, *ts):
        self.saved_tensors = ts

    def mark_non_differentiable(self, *a):
        pass

    def set_materialize_grads(self, v):
        pass


def _require_gpu(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"splatloc_amd rasterizer: `{name}` is on {t.device}; tensors must be on a ROCm device "
            "(the HIP kernels are the only implementation, there is no CPU fallback)")


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    # the camera tensors are passed as explicit autograd inputs as well, so that a pose
    # parametrisation upstream of viewmatrix / projmatrix / campos receives gradients
    # (pose-gradient extension; the reference has no such path)
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, raster_settings.viewmatrix,
                                     raster_settings.projmatrix, raster_settings.campos)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings: GaussianRasterizationSettings, viewmatrix=None, projmatrix=None, campos=None):
        lib = _native.load()
        _require_gpu(means3D, "means3D")
        dev = means3D.device
        rs = raster_settings
        P = int(means3D.shape[0])
        H, W = int(rs.image_height), int(rs.image_width)

        m3 = _prep(means3D, dev)
        shs = _prep(sh, dev)
        col = _prep(colors_precomp, dev)
        opa = _prep(opacities, dev)
        sca = _prep(scales, dev)
        rot = _prep(rotations, dev)
        cov = _prep(cov3Ds_precomp, dev)
        bg = _prep(rs.bg, dev)
        view = _prep(rs.viewmatrix if viewmatrix is None else viewmatrix, dev)
        proj = _prep(rs.projmatrix if projmatrix is None else projmatrix, dev)
        campos = _prep(rs.campos if campos is None else campos, dev)

        if shs is not None:
            Cn, M = 3, int(shs.shape[1])
        elif col is not None:
            Cn, M = int(col.shape[1]), 0
        elif colors_precomp is not None and colors_precomp.dim() == 2 and colors_precomp.shape[1] > 0:
            Cn, M = int(colors_precomp.shape[1]), 0     # P = 0: the channel count is still the table's width
        else:
            Cn, M = (3, 0)
        st = _native.Settings(H, W, float(rs.tanfovx), float(rs.tanfovy), float(rs.scale_modifier),
                              int(rs.sh_degree), M, Cn, 0 if bg is None else int(bg.numel()),
                              int(bool(rs.prefiltered)), int(bool(rs.debug)))

        color = torch.empty((Cn, H, W), dtype=torch.float32, device=dev)
        depth = torch.empty((1, H, W), dtype=torch.float32, device=dev)
        alpha = torch.empty((1, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)   # preprocess_kernel writes every element
        geom = torch.empty((lib.splatraster_geometry_bytes(P),), dtype=torch.uint8, device=dev)
        img = torch.empty((lib.splatraster_image_bytes(W, H),), dtype=torch.uint8, device=dev)
        stream = _stream(dev)
        R = C.c_int64(0)
        with _on_device(dev):
            _native.check(lib.splatraster_forward_geometry(
                C.byref(st), P, _ptr(m3), _ptr(shs), _ptr(opa), _ptr(sca), _ptr(rot), _ptr(cov), _ptr(view),
                _ptr(proj), _ptr(campos), _ptr(geom), _ptr(radii), C.byref(R), stream), "forward_geometry")
            binning = torch.empty((lib.splatraster_binning_bytes(P, R.value, W, H, Cn),), dtype=torch.uint8,
                                  device=dev)
            _native.check(lib.splatraster_forward_render(
                C.byref(st), P, R.value, _ptr(bg), _ptr(col), _ptr(geom), _ptr(binning), _ptr(img),
                _ptr(color), _ptr(depth), _ptr(alpha), stream), "forward_render")

        ctx.raster_settings = rs
        ctx.st = st
        ctx.num_rendered = int(R.value)
        ctx.shapes = (tuple(means3D.shape), None if sh is None else tuple(sh.shape))
        ctx.have = (shs is not None, col is not None, sca is not None, cov is not None)
        none = _empty(dev)
        ctx.save_for_backward(*[t if t is not None else none for t in
                                (m3, shs, col, opa, sca, rot, cov, bg, view, proj, campos)],
                              radii, geom, binning, img, color, depth, alpha)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)   # no zero tensors for unused output gradients (radii, depth, alpha)
        return color, depth, alpha, radii

    @staticmethod
    def backward(ctx, grad_color, grad_depth, grad_alpha, _grad_radii=None):
        lib = _native.load()
        (m3, shs, col, opa, sca, rot, cov, bg, view, proj, campos, radii, geom, binning, img, color, depth,
         alpha) = ctx.saved_tensors
        dev = m3.device
        opt = lambda t: t if t.numel() else None  # noqa: E731
        shs, col, sca, rot, cov, bg, campos = map(opt, (shs, col, sca, rot, cov, bg, campos))
        st = ctx.st
        P = int(m3.shape[0])
        Cn = st.channels
        g_color = _prep(grad_color, dev)
        if g_color is None:
            g_color = torch.zeros_like(color)
        g_depth = _prep(grad_depth, dev) if grad_depth is not None else None
        g_alpha = _prep(grad_alpha, dev) if grad_alpha is not None else None

        f32 = dict(dtype=torch.float32, device=dev)
        # every parameter gradient of the frame is carved out of ONE allocation (16-byte
        # aligned pieces), so a frame-parallel replica can all-reduce them where they are as a
        # single RCCL call (frame_parallel.allreduce_grads); the viewspace gradient is per-view
        # state and stays outside
        shapes = {"m3": (P, 3), "op": (P, 1)}
        if col is not None:
            shapes["col"] = (P, Cn)
        if sca is not None:
            shapes["sca"] = (P, 3)
        if rot is not None:
            shapes["rot"] = (P, 4)
        if cov is not None:
            shapes["cov"] = (P, 6)
        if shs is not None:
            shapes["sh"] = tuple(shs.shape)
        offs, total = {}, 0
        for k, shp in shapes.items():
            offs[k] = total
            n = 1
            for d in shp:
                n *= int(d)
            total += (n + 3) & ~3
        flat = torch.empty((total,), **f32)

        def piece(k):
            if k not in shapes:
                return None
            n = 1
            for d in shapes[k]:
                n *= int(d)
            return flat[offs[k]:offs[k] + n].view(shapes[k])

        d_m3, d_op, d_col, d_sca, d_rot, d_cov, d_sh = (piece(k) for k in ("m3", "op", "col", "sca", "rot", "cov", "sh"))
        d_m2 = torch.empty((P, 3), **f32)
        want_pose = any(ctx.needs_input_grad[9:12]) if len(ctx.needs_input_grad) >= 12 else False
        d_view = torch.empty((4, 4), **f32) if want_pose else None
        d_proj = torch.empty((4, 4), **f32) if want_pose else None
        d_cam = torch.empty((3,), **f32) if (want_pose and campos is not None) else None   # (written in full by the backward)
        with _on_device(dev):
            _native.check(lib.splatraster_backward(
                C.byref(st), P, ctx.num_rendered, _ptr(bg), _ptr(m3), _ptr(shs), _ptr(col), _ptr(opa),
                _ptr(sca), _ptr(rot), _ptr(cov), _ptr(view), _ptr(proj), _ptr(campos), _ptr(radii),
                _ptr(geom), _ptr(binning), _ptr(img), _ptr(color), _ptr(depth), _ptr(alpha), _ptr(g_color),
                _ptr(g_depth), _ptr(g_alpha), _ptr(d_m3), _ptr(d_m2), _ptr(d_col), _ptr(d_op), _ptr(d_sca),
                _ptr(d_rot), _ptr(d_cov), _ptr(d_sh), _ptr(d_view), _ptr(d_proj), _ptr(d_cam), _stream(dev)),
                "backward")
        # (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, settings,
        #  viewmatrix, projmatrix, campos)
        return d_m3, d_m2, d_sh, d_col, d_op, d_sca, d_rot, d_cov, None, d_view, d_proj, d_cam


def window_grad_layout(P: int, Cn: int, have_scales: bool = True, have_cov: bool = False):
    """Layout of the ONE allocation that holds the summed parameter gradients of a window: 16-byte aligned pieces
    `m3 [P,3] | op [P,1] | col [P,Cn] | sca [P,3] | rot [P,4] | cov [P,6]`, optionally followed by a `[2, P]` TAIL for the
    increments of xyz_gradient_accum / denom, so that a frame-parallel replica reduces gradients AND statistics as one
    in-place SUM (frame_parallel.reduce_step).  Returns (shapes, offsets, total_floats)."""
    shapes = {"m3": (P, 3), "op": (P, 1), "col": (P, Cn)}
    if have_scales:
        shapes["sca"], shapes["rot"] = (P, 3), (P, 4)
    if have_cov:
        shapes["cov"] = (P, 6)
    offs, total = {}, 0
    for k, shp in shapes.items():
        offs[k] = total
        total += (shp[0] * shp[1] + 3) & ~3
    return shapes, offs, total


def window_grad_span(P: int, Cn: int, device, have_scales: bool = True, have_cov: bool = False, tail: bool = True,
                     zero: bool = False) -> dict:
    """Allocates the gradient allocation of `window_grad_layout` (+ the statistics tail).  `zero=True`: what a rank
    WITHOUT views in a frame-parallel step contributes — zero gradients in exactly the layout the other ranks' backward
    produced, so that every rank reduces the same buffer.  Keys: flat, tail ([2, P, 1] or None), m3, op, col, sca, rot, cov."""
    shapes, offs, total = window_grad_layout(P, Cn, have_scales, have_cov)
    n = total + (2 * P if tail else 0)
    flat = (torch.zeros if zero else torch.empty)((n,), dtype=torch.float32, device=device)
    out = {"flat": flat, "tail": None}
    for k in ("m3", "op", "col", "sca", "rot", "cov"):
        out[k] = flat[offs[k]:offs[k] + shapes[k][0] * shapes[k][1]].view(shapes[k]) if k in shapes else None
    if tail:
        out["tail"] = flat[total:total + 2 * P].view(2, P, 1)
        if not zero:
            out["tail"].zero_()
    return out


def _window_compatible(settings) -> bool:
    """One launch sequence needs one image size, channel layout, scale modifier and background for all views."""
    if not settings:
        return True
    a = settings[0]
    for b in settings[1:]:
        if (int(b.image_height), int(b.image_width)) != (int(a.image_height), int(a.image_width)):
            return False
        if float(b.scale_modifier) != float(a.scale_modifier):
            return False
        if b.bg is not a.bg and b.bg.data_ptr() != a.bg.data_ptr() and not torch.equal(b.bg, a.bg):   # (the last test reads the device: only for distinct tensors)
            return False
    return True


class _RasterizeWindow(torch.autograd.Function):
    """The V views of one optimisation window (train_gaussians.py:195-229) as ONE launch sequence:
    splatraster_forward_window_* / splatraster_backward_window (include/splatraster.h).  Inputs: the shared
    rasterizer arguments, the list of per-view settings, then one `means2D` gradient carrier per view.  Outputs:
    (color_0, depth_0, alpha_0, radii_0, color_1, ...).  The backward runs once, when autograd has the output
    gradients of every view, and returns parameter gradients already summed over the views.

    `split_last`: the colour buffer of every view is handed out as TWO autograd outputs, channels [0, C-1) and channel
    C-1 — SplatLoc's `render` = image[:3] and `kp_prob` = image[-1] (gaussian_renderer/__init__.py:133-135) — so the
    outputs are (rgb_0, last_0, depth_0, alpha_0, radii_0, rgb_1, ...).  Their gradients then reach the kernel as
    separate planes (no zero-padded [C,H,W] copies and no add, which is what slicing one output costs in autograd),
    and a channel / auxiliary plane that did not reach the loss is skipped by the backward (color_refinement).
    `split_last` may also be an integer g in [1, C - 1): the outputs are then (image[:g], image[-1], ...) and the channels
    in between are not handed out at all (a wide [rgb | features | kp_score] table whose loss reads rgb and kp_score)."""

    @staticmethod
    def forward(ctx, means3D, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, settings, split_last, grad_span,
                *means2D):
        lib = _native.load()
        ctx.grad_span = grad_span
        _require_gpu(means3D, "means3D")
        dev = means3D.device
        V = len(settings)
        assert 1 <= V <= _native.MAX_WINDOW_VIEWS and len(means2D) == V
        rs0 = settings[0]
        P = int(means3D.shape[0])
        H, W = int(rs0.image_height), int(rs0.image_width)
        m3, col, opa = _prep(means3D, dev), _prep(colors_precomp, dev), _prep(opacities, dev)
        sca, rot, cov = _prep(scales, dev), _prep(rotations, dev), _prep(cov3Ds_precomp, dev)
        bg = _prep(rs0.bg, dev)
        Cn = int(colors_precomp.shape[1])
        st = _native.Settings(H, W, float(rs0.tanfovx), float(rs0.tanfovy), float(rs0.scale_modifier), 0, 0, Cn,
                              0 if bg is None else int(bg.numel()), 0, 0)
        f32 = dict(dtype=torch.float32, device=dev)
        cams = [(_prep(rs.viewmatrix, dev), _prep(rs.projmatrix, dev), _prep(rs.campos, dev)) for rs in settings]
        # one allocation per kind; the per-view outputs are its slices (plain tensors for autograd: they do not
        # alias any input)
        color = torch.empty((V, Cn, H, W), **f32)
        depth = torch.empty((V, 1, H, W), **f32)
        alpha = torch.empty((V, 1, H, W), **f32)
        radii = torch.empty((V, P), dtype=torch.int32, device=dev)
        views = (_native.WindowView * V)()
        for v, rs in enumerate(settings):
            w = views[v]
            w.viewmatrix, w.projmatrix = cams[v][0].data_ptr(), cams[v][1].data_ptr()
            w.campos = None if cams[v][2] is None else cams[v][2].data_ptr()
            w.tanfovx, w.tanfovy = float(rs.tanfovx), float(rs.tanfovy)
            w.radii = radii[v].data_ptr() if P else None
            w.out_color, w.out_depth, w.out_alpha = color[v].data_ptr(), depth[v].data_ptr(), alpha[v].data_ptr()
        geom = torch.empty((lib.splatraster_window_geometry_bytes(P, V),), dtype=torch.uint8, device=dev)
        img = torch.empty((lib.splatraster_window_image_bytes(W, H, V),), dtype=torch.uint8, device=dev)
        stream = _stream(dev)
        R = (C.c_int64 * V)()
        raw_fwd = getattr(ctx, "raw_fwd", None)
        with _on_device(dev):
            if raw_fwd is not None:
                # RAW-parameter mode (training's graph-free paths): colors_precomp / opacities / scales / rotations are EMPTY tensors that
                # the projection kernel fills from the raw parameters (ctx.raw_fwd = (scaling, rotation, opacity, f_dc, extra or None))
                sc_r, ro_r, op_r, fd_r, ex_r = raw_fwd
                rf = _native.RawForward()
                rf.scaling, rf.rotation, rf.opacity, rf.f_dc = sc_r.data_ptr(), ro_r.data_ptr(), op_r.data_ptr(), fd_r.data_ptr()
                rf.extra = None if ex_r is None else ex_r.data_ptr()
                rf.extra_channels = 0 if ex_r is None else int(ex_r.shape[1])
                rf.scales, rf.rotations, rf.opacities, rf.colors = sca.data_ptr(), rot.data_ptr(), opa.data_ptr(), col.data_ptr()
                _native.check(lib.splatraster_forward_window_geometry_raw(
                    C.byref(st), V, views, P, _ptr(m3), C.byref(rf), _ptr(geom), R, stream), "forward_window_geometry_raw")
            else:
                _native.check(lib.splatraster_forward_window_geometry(
                    C.byref(st), V, views, P, _ptr(m3), _ptr(opa), _ptr(sca), _ptr(rot), _ptr(cov), _ptr(geom), R, stream),
                    "forward_window_geometry")
            Rt = sum(int(r) for r in R)
            binning = torch.empty((lib.splatraster_window_binning_bytes(P, V, Rt, W, H, Cn),), dtype=torch.uint8,
                                  device=dev)
            _native.check(lib.splatraster_forward_window_render(
                C.byref(st), V, views, P, R, _ptr(bg), _ptr(col), _ptr(geom), _ptr(binning), _ptr(img), stream),
                "forward_window_render")
        head = 0
        if split_last is True:
            head = Cn - 1
        elif not isinstance(split_last, bool) and split_last:
            head = int(split_last)
        split_last = Cn >= 2 and 1 <= head <= Cn - 1
        ctx.split_last = split_last
        ctx.head = head if split_last else 0
        ctx.st, ctx.V, ctx.R = st, V, [int(r) for r in R]
        ctx.tanfov = [(float(rs.tanfovx), float(rs.tanfovy)) for rs in settings]
        ctx.have = (sca is not None, cov is not None)
        none = _empty(dev)
        flat_cams = [t if t is not None else none for cam in cams for t in cam]
        # the background is an INPUT of the backward: the back-to-front walk starts every pixel at A = bg . g - g_A (DESIGN.md
        # §6.3), so it is saved like a tensor the backward reads (autograd's version check then catches an in-place change
        # between forward and backward) — not kept as a bare attribute
        ctx.have_bg = bg is not None
        ctx.save_for_backward(*[t if t is not None else none for t in (m3, col, opa, sca, rot, cov)], radii, geom, binning,
                              img, color, depth, alpha, *flat_cams, bg if bg is not None else none)
        outs, rad = [], []
        for v in range(V):
            rv = radii[v]
            rad.append(rv)
            if split_last:
                outs += [color[v, :head], color[v, Cn - 1], depth[v], alpha[v], rv]
            else:
                outs += [color[v], depth[v], alpha[v], rv]
        ctx.mark_non_differentiable(*rad)
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        lib = _native.load()
        saved = ctx.saved_tensors
        m3, col, opa, sca, rot, cov, radii, geom, binning, img, color, depth, alpha = saved[:13]
        flat_cams = saved[13:-1]
        bg = saved[-1] if ctx.have_bg else None
        dev = m3.device
        V, st = ctx.V, ctx.st
        opt = lambda t: t if t.numel() else None  # noqa: E731
        sca, rot, cov = map(opt, (sca, rot, cov))
        P, Cn = int(m3.shape[0]), st.channels
        f32 = dict(dtype=torch.float32, device=dev)
        # the summed parameter gradients of the window: 16-byte aligned pieces of ONE allocation, like the per-view
        # call, so a frame-parallel replica all-reduces them in place as a single RCCL call; with `grad_span` the
        # allocation ends with a zeroed [2, P] tail for the statistics increments, which then ride in the same call
        span = window_grad_span(P, Cn, dev, have_scales=sca is not None, have_cov=cov is not None,
                                tail=ctx.grad_span is not None)
        if ctx.grad_span is not None:
            # only the whole allocation and its tail: holding the PIECES here would raise their use count and autograd's
            # AccumulateGrad would then deep-copy them instead of keeping them as the parameters' .grad
            ctx.grad_span.append({"flat": span["flat"], "tail": span["tail"]})
        d_m3, d_op, d_col, d_sca, d_rot, d_cov = (span[k] for k in ("m3", "op", "col", "sca", "rot", "cov"))
        d_m2 = torch.empty((V, P, 3), **f32)
        views = (_native.WindowView * V)()
        keep = []
        zeros_color = None
        split = ctx.split_last
        nout = 5 if split else 4
        for v in range(V):
            if split:
                g_color, g_last, g_depth, g_alpha = gouts[nout * v:nout * v + 4]
                g_last = _prep(g_last, dev) if g_last is not None else None
            else:
                (g_color, g_depth, g_alpha), g_last = gouts[nout * v:nout * v + 3], None
            g_color = _prep(g_color, dev) if g_color is not None else None
            if g_color is None:     # this view's colour buffer did not reach the loss
                if zeros_color is None:
                    zeros_color = torch.zeros((Cn, st.image_height, st.image_width), **f32)
                g_color = zeros_color
            g_depth = _prep(g_depth, dev) if g_depth is not None else None
            g_alpha = _prep(g_alpha, dev) if g_alpha is not None else None
            keep += [g_color, g_depth, g_alpha, g_last]
            w = views[v]
            cv, cp, cc = flat_cams[3 * v], flat_cams[3 * v + 1], flat_cams[3 * v + 2]
            w.viewmatrix, w.projmatrix = cv.data_ptr(), cp.data_ptr()
            w.campos = cc.data_ptr() if cc.numel() else None
            w.tanfovx, w.tanfovy = ctx.tanfov[v]
            w.radii = radii[v].data_ptr() if P else None
            w.out_color, w.out_depth, w.out_alpha = color[v].data_ptr(), depth[v].data_ptr(), alpha[v].data_ptr()
            w.dL_dout_color = g_color.data_ptr()
            w.dL_dout_depth = None if g_depth is None else g_depth.data_ptr()
            w.dL_dout_alpha = None if g_alpha is None else g_alpha.data_ptr()
            w.dL_dmeans2D = d_m2[v].data_ptr() if P else None
            w.dL_dout_last = None if g_last is None else g_last.data_ptr()
            w.color_grad_channels = ctx.head if split else 0
        R = (C.c_int64 * V)(*ctx.R)
        raw = getattr(ctx, "raw", None)
        if raw is not None:
            # RAW-parameter mode (training's graph-free paths: SplatLoc's own configuration): the chain through the activations and
            # the colour gather run inside the per-Gaussian backward kernel (splatraster_backward_window_raw) — no dL/dcolors /
            # dL/dopacities / dL/dscales / dL/drotations tensors, no activation-backward launch.  ctx.raw = (scaling [P,3],
            # rotation [P,4], opacity [P,1], f_dc [P,1,3], extra [P,E] or None); the gradients are left in ctx.raw_out.
            sc_r, ro_r, op_r, fd_r, ex_r = raw[:5]
            reg = raw[5] if len(raw) > 5 else None       # (row_grad [P], out [2], weight): the isotropic regulariser's term (map step)
            E = 0 if ex_r is None else int(ex_r.shape[1])
            d_sc, d_ro, d_opr = torch.empty((P, 3), **f32), torch.empty((P, 4), **f32), torch.empty((P, 1), **f32)
            d_fd = torch.empty(tuple(fd_r.shape), **f32)
            d_ex = torch.empty((P, E), **f32) if E else None
            rp = _native.RawParams()
            rp.scaling, rp.rotation, rp.opacity, rp.f_dc = sc_r.data_ptr(), ro_r.data_ptr(), op_r.data_ptr(), fd_r.data_ptr()
            rp.extra_channels = E
            rp.dL_dscaling, rp.dL_drotation, rp.dL_dopacity, rp.dL_df_dc = d_sc.data_ptr(), d_ro.data_ptr(), d_opr.data_ptr(), d_fd.data_ptr()
            rp.dL_dextra = d_ex.data_ptr() if d_ex is not None else None
            if reg is not None:
                rp.reg_row_grad, rp.reg_out, rp.reg_weight = reg[0].data_ptr(), reg[1].data_ptr(), float(reg[2])
            with _on_device(dev):
                _native.check(lib.splatraster_backward_window_raw(
                    C.byref(st), V, views, P, R, _ptr(bg), _ptr(m3), _ptr(col), _ptr(sca), _ptr(rot), _ptr(geom), _ptr(binning),
                    _ptr(img), C.byref(rp), _ptr(d_m3), _stream(dev)), "backward_window_raw")
            del keep
            ctx.raw_out = (d_sc, d_ro, d_opr, d_fd, d_ex)
            return (d_m3, None, None, None, None, None, None, None, None) + tuple(d_m2[v] for v in range(V))
        with _on_device(dev):
            _native.check(lib.splatraster_backward_window(
                C.byref(st), V, views, P, R, _ptr(bg), _ptr(m3), _ptr(col), _ptr(sca), _ptr(rot), _ptr(cov), _ptr(geom),
                _ptr(binning), _ptr(img), _ptr(d_m3), _ptr(d_col), _ptr(d_op), _ptr(d_sca), _ptr(d_rot), _ptr(d_cov),
                _stream(dev)), "backward_window")
        del keep
        # (means3D, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, settings, split_last, grad_span, *means2D)
        return (d_m3, d_col, d_op, d_sca, d_rot, d_cov, None, None, None) + tuple(d_m2[v] for v in range(V))


def rasterize_window(settings, means3D, means2D, colors_precomp, opacities, scales=None, rotations=None,
                     cov3D_precomp=None, split_last: bool = False, grad_span: Optional[list] = None):
    """`[GaussianRasterizer(s)(means3D, m2, opacities, colors_precomp=..., ...) for s, m2 in zip(settings, means2D)]`
    as one launch sequence per chunk of <= 8 views.  `settings`: GaussianRasterizationSettings per view (same
    image size / scale modifier / background); `means2D`: one gradient carrier per view.  Returns a list of
    (color, depth, alpha, radii) per view — bit-identical to the per-view calls; the backward sums the views'
    parameter gradients in-kernel (one gradient set per window instead of V sets + V accumulation passes).
    `split_last`: (rgb [C-1,H,W], last [H,W], depth, alpha, radii) per view instead (an integer g: (image[:g], image[-1],
    ...)) — see _RasterizeWindow.
    `grad_span`: a list; every chunk's backward appends its gradient allocation (`window_grad_span`: flat, the pieces and a
    zeroed [2, P, 1] tail for the xyz_gradient_accum / denom increments) — the frame-parallel step reduces gradients and
    statistics as ONE in-place SUM (frame_parallel.reduce_step).
    An empty window (no settings) returns []."""
    settings, means2D = list(settings), list(means2D)
    if not settings and not means2D:
        return []
    if colors_precomp is None:
        raise Exception("rasterize_window needs precomputed colors (view-dependent SH colours: per-view calls)")
    if ((scales is None or rotations is None) and cov3D_precomp is None) or (
            (scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
    if len(settings) != len(means2D) or not settings:
        raise Exception("rasterize_window: one means2D tensor per view")
    if not _window_compatible(settings):
        raise Exception("rasterize_window: the views of a window share image size, scale modifier and background")
    out = []
    # chunks of at most 8 views, and of at most 2^24 (view, Gaussian) rows: the kernels address rows with 24-bit multiplies
    # and the depth-order words keep the row in 24 bits (a scene of > 2 M Gaussians renders its window in smaller chunks)
    K = max(1, min(_native.MAX_WINDOW_VIEWS, (1 << 24) // max(int(means3D.shape[0]), 1)))
    for a in range(0, len(settings), K):
        flat = _RasterizeWindow.apply(means3D, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                      tuple(settings[a:a + K]), split_last, grad_span, *means2D[a:a + K])
        Cn = int(colors_precomp.shape[1])
        head = Cn - 1 if split_last is True else (int(split_last) if split_last else 0)
        n = 5 if (Cn >= 2 and 1 <= head <= Cn - 1) else 4
        out += [tuple(flat[n * v:n * v + n]) for v in range(len(flat) // n)]
    return out


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """Boolean mask of points in front of the near plane (view z > 0.2)."""
        lib = _native.load()
        _require_gpu(positions, "positions")
        rs = self.raster_settings
        dev = positions.device
        with torch.no_grad():
            pos = _prep(positions, dev)
            P = int(positions.shape[0])
            present = torch.zeros((P,), dtype=torch.uint8, device=dev)
            if P:
                view, proj = _prep(rs.viewmatrix, dev), _prep(rs.projmatrix, dev)
                with torch.cuda.device(dev):
                    _native.check(lib.splatraster_mark_visible(P, _ptr(pos), _ptr(view), _ptr(proj),
                                                               _ptr(present), _stream(dev)), "mark_visible")
        return present.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, rs)
