"""Pose parameterisations and a pose-refinement loop on the rasterizer's pose gradients.

The reference ships `utils/optimization_utils.py` (axis-angle / quaternion / 6-D rotation + translation -> 4x4 transform) for
optimising a camera pose through the renderer, but nothing in the reference calls it and its rasterizer returns no gradient
for the camera: the path ends at `viewmatrix`.  Here it continues — `diff_gauss.GaussianRasterizer` returns
dL/dviewmatrix, dL/dprojmatrix and dL/dcampos when those tensors require a gradient (tests/test_gpu_pose.py) —
so the helpers have a use.  Same names, argument meaning and return shapes as utils/optimization_utils.py:5-66, restated
without pytorch3d (not installed here); the rotation conversions follow the published formulas pytorch3d implements
(quaternions real part first; 6-D rotations: Zhou et al., "On the Continuity of Rotation Representations", rows b1, b2, b1 x b2).

Two deliberate differences, both where the reference is broken:
  * `axis_angle_to_matrix` of the ZERO vector is the identity here (the reference divides by |w| = 0 and returns NaN —
    its own "TODO: Identity would cause the problem", optimization_utils.py:4); gradients at zero are those of the
    first-order expansion R = I + [w]x;
  * `six_t_to_transform_matrix` returns the matrix (the reference's last line is a bare `return`, :66: it returns None).

Host-side torch code: these are 3x3 / 4x4 operations per camera, not a kernel.
"""
from __future__ import annotations

import torch


def _skew(v: torch.Tensor) -> torch.Tensor:
    z = torch.zeros_like(v[..., 0])
    return torch.stack([torch.stack([z, -v[..., 2], v[..., 1]], dim=-1),
                        torch.stack([v[..., 2], z, -v[..., 0]], dim=-1),
                        torch.stack([-v[..., 1], v[..., 0], z], dim=-1)], dim=-2)


def axis_angle_to_matrix(data: torch.Tensor) -> torch.Tensor:
    """Rodrigues: [..., 3] axis * angle -> [..., 3, 3] (optimization_utils.py:5-22).  R = I + sin(t) K + (1 - cos(t)) K^2 with
    K = [w / t]x, written as I + a [w]x + b [w]x^2 with a = sin(t) / t, b = (1 - cos(t)) / t^2 so that t -> 0 is regular."""
    t2 = (data * data).sum(dim=-1, keepdim=True)
    small = t2 < 1e-12
    t2s = torch.where(small, torch.ones_like(t2), t2)
    t = torch.sqrt(t2s)
    a = torch.where(small, 1.0 - t2 / 6.0, torch.sin(t) / t)[..., None]
    b = torch.where(small, 0.5 - t2 / 24.0, (1.0 - torch.cos(t)) / t2s)[..., None]
    W = _skew(data)
    eye = torch.eye(3, dtype=data.dtype, device=data.device).expand(*data.shape[:-1], 3, 3)
    return eye + a * W + b * (W @ W)


def quaternion_to_matrix(q: torch.Tensor) -> torch.Tensor:
    """[..., 4] quaternion (real part first, any non-zero norm) -> [..., 3, 3]."""
    r, i, j, k = q.unbind(-1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack([1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)], dim=-1)
    return o.reshape(q.shape[:-1] + (3, 3))


def matrix_to_quaternion(R: torch.Tensor) -> torch.Tensor:
    """[..., 3, 3] rotation -> [..., 4] unit quaternion, real part first and non-negative.  The candidate with the largest
    of (1 + trace, 1 + 2 R_ii - trace) is used: its square root is >= 1/2, so no component is divided by a small number."""
    m = R.reshape(R.shape[:-2] + (9,))
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = m.unbind(-1)
    q_abs2 = torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22], dim=-1)
    q_abs = torch.sqrt(q_abs2.clamp_min(0.0))
    cand = torch.stack([
        torch.stack([q_abs2[..., 0], m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs2[..., 1], m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs2[..., 2], m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs2[..., 3]], dim=-1)], dim=-2)
    cand = cand / (2.0 * q_abs[..., None].clamp_min(0.1))
    best = q_abs.argmax(dim=-1)
    q = torch.gather(cand, -2, best[..., None, None].expand(best.shape + (1, 4))).squeeze(-2)
    return torch.where(q[..., :1] < 0, -q, q)


def quaternion_to_axis_angle(q: torch.Tensor) -> torch.Tensor:
    """[..., 4] unit quaternion (real part first) -> [..., 3] axis * angle, angle in [0, pi] for a non-negative real part."""
    n = torch.linalg.norm(q[..., 1:], dim=-1, keepdim=True)
    r = q[..., :1]
    small = n < 1e-6
    rs = torch.where(small, r.clamp_min(1e-6), torch.ones_like(r))
    x2 = (n / rs) ** 2
    # angle / |v| = 2 atan2(|v|, r) / |v|; for |v| -> 0: (2 / r) (1 - (|v| / r)^2 / 3)
    scale = torch.where(small, (2.0 / rs) * (1.0 - x2 / 3.0), 2.0 * torch.atan2(n, r) / torch.where(small, torch.ones_like(n), n))
    return q[..., 1:] * scale


def matrix_to_axis_angle(rot: torch.Tensor) -> torch.Tensor:
    """[N, 3, 3] -> [N, 3] (optimization_utils.py:24-29)."""
    return quaternion_to_axis_angle(matrix_to_quaternion(rot))


def rotation_6d_to_matrix(d6: torch.Tensor) -> torch.Tensor:
    """[..., 6] -> [..., 3, 3]: Gram-Schmidt of the two 3-vectors; ROWS b1, b2, b1 x b2."""
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = torch.nn.functional.normalize(a1, dim=-1)
    b2 = torch.nn.functional.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1, dim=-1)
    return torch.stack((b1, b2, torch.cross(b1, b2, dim=-1)), dim=-2)


def _transform(R: torch.Tensor, trans: torch.Tensor) -> torch.Tensor:
    bs = R.shape[0]
    bottom = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=R.dtype, device=R.device).expand(bs, 1, 4)
    return torch.cat([torch.cat([R, trans[:, :, None]], dim=2), bottom], dim=1)


def at_to_transform_matrix(rot: torch.Tensor, trans: torch.Tensor) -> torch.Tensor:
    """axis-angle [bs, 3] + translation [bs, 3] -> [bs, 4, 4] (optimization_utils.py:31-42)."""
    return _transform(axis_angle_to_matrix(rot), trans)


def qt_to_transform_matrix(rot: torch.Tensor, trans: torch.Tensor) -> torch.Tensor:
    """quaternion [bs, 4] (real part first) + translation [bs, 3] -> [bs, 4, 4] (optimization_utils.py:44-54)."""
    return _transform(quaternion_to_matrix(rot), trans)


def six_t_to_transform_matrix(rot: torch.Tensor, trans: torch.Tensor) -> torch.Tensor:
    """6-D rotation [bs, 6] + translation [bs, 3] -> [bs, 4, 4] (optimization_utils.py:55-66; see the module header)."""
    return _transform(rotation_6d_to_matrix(rot), trans)


def camera_tensors(W2C: torch.Tensor, projection_matrix: torch.Tensor):
    """(world_view_transform, full_proj_transform, camera_center) as differentiable functions of a [4, 4] world-to-camera
    matrix — utils/camera_utils.py:129-139: the row-vector (transposed) view matrix, view @ projection, and the camera centre
    -R^T t (the reference inverts the 4x4; for a rigid transform that is the same point, without the LU)."""
    view = W2C.transpose(0, 1)
    proj = view @ projection_matrix
    campos = -(W2C[:3, :3].transpose(0, 1) @ W2C[:3, 3])
    return view, proj, campos


def refine_pose(render_target, gaussians: dict, camera, W2C_init: torch.Tensor, iterations: int = 100, lr_rot: float = 2e-3,
                lr_trans: float = 3e-3, depth_weight: float = 0.2, background: torch.Tensor | None = None, on_step=None,
                graph_free: bool = True):
    """Gradient descent on a camera pose through the rasterizer: the pose is W2C = T(w, t) @ W2C_init with an axis-angle w
    and a translation t (both start at zero), the loss is L1(colour) + depth_weight * L1(depth) against `render_target` =
    (colour [C,H,W], depth [1,H,W] or None), Adam on (w, t).  `gaussians`: dict(means3D, colors, opacities, scales, rotations)
    of device tensors (activated values, as the rasterizer takes them); `camera`: intrinsics holder with image_width / height,
    tanfovx / tanfovy and `projection_matrix` (splatloc_amd.camera.PinholeCamera or the reference's Camera).
    Returns (W2C [4,4] detached, history of loss values as one device tensor [iterations]).

    `graph_free` (default): an iteration is ONE launch sequence with no torch operator in it — the rasterizer's forward and
    backward called directly (`PlainCtx`, like training.color_refinement_step), `splatraster_l1_rgbd_loss` for the loss and
    its gradient planes, `splatraster_pose_step` for the chain rule to (w, t), the Adam step and the next camera tensors
    (csrc/pose.hip): no autograd graph, no torch.optim, the 6 numbers never visit the host.  `graph_free=False` is round
    4's loop (autograd through the 4x4 algebra + torch.optim.Adam): the same iterates to float32 rounding
    (tests/test_gpu_pose.py), ~3x the host time per iteration."""
    if not graph_free:
        return _refine_pose_autograd(render_target, gaussians, camera, W2C_init, iterations, lr_rot, lr_trans, depth_weight,
                                     background, on_step)
    import ctypes as C
    from . import _native
    from .rasterizer import GaussianRasterizationSettings, PlainCtx, _RasterizeGaussians, _stream
    lib = _native.load()
    dev = gaussians["means3D"].device
    tgt_c, tgt_d = render_target
    tgt_c = tgt_c.to(dev).float().contiguous()
    tgt_d = None if (tgt_d is None or not depth_weight) else tgt_d.to(dev).float().contiguous()
    W2C0 = W2C_init.to(dev).float().contiguous()
    Pm = camera.projection_matrix.to(dev).float().contiguous()
    H, W = int(camera.image_height), int(camera.image_width)
    bg = background if background is not None else torch.zeros(int(tgt_c.shape[0]) if tgt_c.shape[0] <= 3 else 0, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    state = torch.zeros(20, **f32)          # w, t, Adam moments, step (splatraster_pose_step)
    view, proj, campos = torch.empty(4, 4, **f32), torch.empty(4, 4, **f32), torch.empty(3, **f32)
    hist = torch.zeros(iterations, **f32)
    g_color, g_depth = torch.empty_like(tgt_c), torch.empty((1, H, W), **f32)
    carrier = torch.zeros_like(gaussians["means3D"])
    ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
    b1, b2, eps = 0.9, 0.999, 1e-8          # torch.optim.Adam's defaults, as round 4's loop used them

    def step(advance, grads=(None, None, None)):
        _native.check(lib.splatraster_pose_step(ptr(grads[0]), ptr(grads[1]), ptr(grads[2]), ptr(W2C0), ptr(Pm), lr_rot, lr_trans,
                                                b1, b2, eps, advance, ptr(state), ptr(view), ptr(proj), ptr(campos), _stream(dev)),
                      "pose_step")

    step(0)
    rs = GaussianRasterizationSettings(H, W, camera.tanfovx, camera.tanfovy, bg, 1.0, view, proj, 0, campos, False, False)
    n_c, n_d = tgt_c.numel(), H * W
    with torch.no_grad():
        for it in range(iterations):
            ctx = PlainCtx()
            color, depth, _alpha, _radii = _RasterizeGaussians.forward(
                ctx, gaussians["means3D"], carrier, None, gaussians["colors"], gaussians["opacities"], gaussians["scales"],
                gaussians["rotations"], None, rs, view, proj, campos)
            _native.check(lib.splatraster_l1_rgbd_loss(n_c, ptr(color), ptr(tgt_c), n_d, ptr(depth), ptr(tgt_d), float(depth_weight),
                                                       ptr(g_color), ptr(g_depth), C.c_void_p(hist.data_ptr() + 4 * it),
                                                       _stream(dev)), "l1_rgbd_loss")
            grads = _RasterizeGaussians.backward(ctx, g_color, g_depth if tgt_d is not None else None, None)
            step(1, grads[9:12])
            if on_step:
                on_step(it, hist[it])
        w, t = state[:3].clone()[None], state[3:6].clone()[None]
        return (at_to_transform_matrix(w, t)[0] @ W2C0).detach(), hist


def _refine_pose_autograd(render_target, gaussians: dict, camera, W2C_init: torch.Tensor, iterations: int = 100,
                          lr_rot: float = 2e-3, lr_trans: float = 3e-3, depth_weight: float = 0.2,
                          background: torch.Tensor | None = None, on_step=None):
    """round 4's refine_pose: autograd through the pose algebra and the drop-in autograd.Function, torch.optim.Adam."""
    from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    dev = gaussians["means3D"].device
    tgt_c, tgt_d = render_target
    W2C0 = W2C_init.to(dev).float()
    P = camera.projection_matrix.to(dev)
    bg = background if background is not None else torch.zeros(int(tgt_c.shape[0]) if tgt_c.shape[0] <= 3 else 0, device=dev)
    w = torch.zeros(1, 3, device=dev, requires_grad=True)
    t = torch.zeros(1, 3, device=dev, requires_grad=True)
    opt = torch.optim.Adam([{"params": [w], "lr": lr_rot}, {"params": [t], "lr": lr_trans}])
    carrier = torch.zeros_like(gaussians["means3D"])
    hist = torch.zeros(iterations, device=dev)
    for it in range(iterations):
        W2C = at_to_transform_matrix(w, t)[0] @ W2C0
        view, proj, campos = camera_tensors(W2C, P)
        rs = GaussianRasterizationSettings(int(camera.image_height), int(camera.image_width), camera.tanfovx, camera.tanfovy,
                                           bg, 1.0, view, proj, 0, campos, False, False)
        color, depth, _alpha, _radii = GaussianRasterizer(raster_settings=rs)(
            means3D=gaussians["means3D"], means2D=carrier, shs=None, colors_precomp=gaussians["colors"],
            opacities=gaussians["opacities"], scales=gaussians["scales"], rotations=gaussians["rotations"], cov3D_precomp=None)
        loss = (color - tgt_c).abs().mean()
        if tgt_d is not None and depth_weight:
            loss = loss + depth_weight * (depth - tgt_d).abs().mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        hist[it] = loss.detach()
        if on_step:
            on_step(it, loss)
    with torch.no_grad():
        return (at_to_transform_matrix(w, t)[0] @ W2C0).detach(), hist
