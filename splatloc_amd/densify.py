"""Device-side pieces of SplatLoc's densification bookkeeping."""
from __future__ import annotations

import torch

from . import _native
from .rasterizer import _ptr, _require_gpu, _stream, _on_device


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
    with _on_device(dev):
        _native.check(_native.load().splatraster_densification_stats(
            P, _ptr(g), _ptr(radii.contiguous()), _ptr(xyz_gradient_accum), _ptr(denom), _ptr(max_radii2D),
            _stream(dev)), "densification_stats")


def add_densification_stats_window(viewspace_grads, radii, xyz_gradient_accum, denom, max_radii2D) -> None:
    """`add_densification_stats` for the views of a window in ONE launch (view order = the reference's loop order,
    train_gaussians.py:238-245).  `xyz_gradient_accum = denom = None`: only `max_radii2D` is updated — the statistics
    line of SplatLoc.color_refinement (train_gaussians.py:293-294); `viewspace_grads` may then be None."""
    import ctypes as C
    radii = list(radii)
    V = len(radii)
    if V == 0:
        return
    dev = radii[0].device
    _require_gpu(radii[0], "radii")
    P = int(radii[0].numel())
    only_max = xyz_gradient_accum is None
    state = ((max_radii2D, "max_radii2D"),) if only_max else ((xyz_gradient_accum, "xyz_gradient_accum"), (denom, "denom"),
                                                             (max_radii2D, "max_radii2D"))
    for t, name in state:
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != P or t.device != dev:
            raise RuntimeError(f"add_densification_stats: `{name}` must be a contiguous float32 tensor of {P} elements on {dev}")
    lib = _native.load()
    K = _native.MAX_WINDOW_VIEWS
    for a in range(0, V, K):
        rs = [r.contiguous() for r in radii[a:a + K]]
        gs = None
        if not only_max:
            gs = []
            for g in viewspace_grads[a:a + K]:
                g = g.detach()
                if g.dtype != torch.float32 or not g.is_contiguous() or tuple(g.shape) != (P, 3):
                    g = g.to(torch.float32).reshape(P, 3).contiguous()
                gs.append(g)
        n = len(rs)
        rp = (C.c_void_p * n)(*[r.data_ptr() for r in rs])
        gp = None if gs is None else (C.c_void_p * n)(*[g.data_ptr() for g in gs])
        with _on_device(dev):
            _native.check(lib.splatraster_densification_stats_window(
                P, n, gp, rp, None if only_max else _ptr(xyz_gradient_accum), None if only_max else _ptr(denom),
                _ptr(max_radii2D), _stream(dev)), "densification_stats_window")


# ---------------------------------------------------------------------------------------------------
# densify / clone / split / prune with optimizer-state surgery (gaussian_model.py:477-675)
# ---------------------------------------------------------------------------------------------------
GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "marker", "kp_score", "scaling", "rotation")
ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
        "marker": "_marker", "kp_score": "_kp_score", "scaling": "_scaling", "rotation": "_rotation"}


def _row_width(t: torch.Tensor) -> int:
    n = 1
    for d in t.shape[1:]:
        n *= int(d)
    return n


def _model_struct(tensors: dict, P: int, widths: dict):
    import ctypes as C
    m = _native.Model()
    m.P, m.f_rest_width, m.marker_width = P, widths["f_rest"], widths["marker"]
    m.kp_width, m.scaling_width = widths["kp_score"], widths["scaling"]
    for k in GROUPS:
        t = tensors.get(k)
        setattr(m, k, None if (t is None or t.numel() == 0) else C.c_void_p(t.data_ptr()))
    return m


def densify_tensors(params: dict, exp_avg: dict, exp_avg_sq: dict, xyz_gradient_accum, denom, max_grad, min_opacity,
                    extent, max_screen_size, percent_dense, primitive_reg, unit_noise=None, seed: int = 0,
                    draw_id: int = 0, return_sources: bool = False):
    """Tensor-level entry.  params / exp_avg / exp_avg_sq: dicts by group name (GROUPS) of contiguous float32
    ROCm tensors [P, ...]; a group without Adam state is missing (or None) in the two moment dicts.
    Returns (new_params, new_exp_avg, new_exp_avg_sq[, source_row, source_kind]) — freshly allocated tensors
    in the reference's row order.  ONE device->host read (the new row count)."""
    import ctypes as C
    lib = _native.load()
    xyz = params["xyz"]
    _require_gpu(xyz, "xyz")
    dev = xyz.device
    P = int(xyz.shape[0])
    for k in GROUPS:
        t = params[k]
        if t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev or int(t.shape[0]) != P:
            raise RuntimeError(f"densify: group `{k}` must be a contiguous float32 tensor with {P} rows on {dev}")
    widths = {k: _row_width(params[k]) for k in GROUPS}
    if (widths["xyz"], widths["f_dc"], widths["opacity"], widths["rotation"]) != (3, 3, 1, 4):
        raise RuntimeError("densify: unexpected parameter shapes")
    src = _model_struct(params, P, widths)
    acc = xyz_gradient_accum.to(torch.float32).contiguous()
    den = denom.to(torch.float32).contiguous()
    ws = torch.empty((lib.splatraster_densify_workspace_bytes(P),), dtype=torch.uint8, device=dev)
    new_P = C.c_int32(0)
    with _on_device(dev):
        _native.check(lib.splatraster_densify_plan(
            C.byref(src), _ptr(acc), _ptr(den), C.c_float(max_grad), C.c_float(min_opacity), C.c_float(extent),
            C.c_float(percent_dense), int(bool(max_screen_size)), int(bool(primitive_reg)), _ptr(ws), C.byref(new_P),
            _stream(dev)), "densify_plan")
    n = int(new_P.value)
    f32 = dict(dtype=torch.float32, device=dev)
    new_params = {k: torch.empty((n,) + tuple(params[k].shape[1:]), **f32) for k in GROUPS}
    has = [k for k in GROUPS if exp_avg.get(k) is not None and exp_avg_sq.get(k) is not None]
    new_m = {k: torch.empty_like(new_params[k]) for k in has}
    new_v = {k: torch.empty_like(new_params[k]) for k in has}
    srow = torch.empty((n,), dtype=torch.int32, device=dev) if return_sources else None
    skind = torch.empty((n,), dtype=torch.int32, device=dev) if return_sources else None
    noise = None
    if unit_noise is not None:
        noise = unit_noise.to(device=dev, dtype=torch.float32).contiguous()
        if tuple(noise.shape) != (2, P, 3):
            raise RuntimeError("densify: unit_noise must be [2, P, 3]")
    if P and n:
        m_in = _model_struct({k: exp_avg[k].contiguous() for k in has}, P, widths) if has else None
        v_in = _model_struct({k: exp_avg_sq[k].contiguous() for k in has}, P, widths) if has else None
        out = _model_struct(new_params, n, widths)
        m_out = _model_struct(new_m, n, widths) if has else None
        v_out = _model_struct(new_v, n, widths) if has else None
        ref = lambda s: None if s is None else C.byref(s)  # noqa: E731
        with _on_device(dev):
            _native.check(lib.splatraster_densify_apply(
                C.byref(src), ref(m_in), ref(v_in), _ptr(noise), C.c_uint64(int(seed)), C.c_uint64(int(draw_id)),
                _ptr(ws), n, C.byref(out), ref(m_out), ref(v_out), _ptr(srow), _ptr(skind), _stream(dev)),
                "densify_apply")
    if return_sources:
        return new_params, new_m, new_v, srow, skind
    return new_params, new_m, new_v


def densify_and_prune(gaussians, max_grad, min_opacity, extent, max_screen_size, *, seed: int = 0, draw_id: int = None,
                      unit_noise=None):
    """Drop-in for `GaussianModel.densify_and_prune(max_grad, min_opacity, extent, max_screen_size)`
    (gaussian_model.py:655-675; call site train_gaussians.py:251-256) on the reference's own model object:
    reads `gaussians._xyz ... _rotation`, `xyz_gradient_accum`, `denom`, `percent_dense`, `primitive_reg` and
    the torch Adam state of `gaussians.optimizer`, and leaves the object as the reference does — new
    `nn.Parameter`s bound to the model AND to the optimizer's groups, `exp_avg` / `exp_avg_sq` re-sized with
    zero rows for new points and the `step` counters carried over, statistics (incl. `max_radii2D`) zeroed.
    `draw_id` (default: an internal counter) distinguishes successive densifications under one `seed`."""
    from torch import nn
    opt = gaussians.optimizer
    groups = {g["name"]: g for g in opt.param_groups}
    params = {k: getattr(gaussians, ATTR[k]).detach().contiguous() for k in GROUPS}
    m, v = {}, {}
    for k in GROUPS:
        st = opt.state.get(groups[k]["params"][0], None)
        if st is not None and "exp_avg" in st:
            m[k], v[k] = st["exp_avg"], st["exp_avg_sq"]
    if draw_id is None:
        draw_id = getattr(gaussians, "_splatloc_draws", 0)
        gaussians._splatloc_draws = draw_id + 1
    new_p, new_m, new_v = densify_tensors(params, m, v, gaussians.xyz_gradient_accum, gaussians.denom, max_grad,
                                          min_opacity, extent, max_screen_size, gaussians.percent_dense,
                                          gaussians.primitive_reg, unit_noise=unit_noise, seed=seed, draw_id=draw_id)
    for k in GROUPS:
        grp = groups[k]
        old = grp["params"][0]
        st = opt.state.pop(old, None)
        p = nn.Parameter(new_p[k].requires_grad_(True))
        grp["params"][0] = p
        if st is not None:
            if k in new_m:
                st["exp_avg"], st["exp_avg_sq"] = new_m[k], new_v[k]
            opt.state[p] = st
        setattr(gaussians, ATTR[k], p)
    n = new_p["xyz"].shape[0]
    dev = new_p["xyz"].device
    gaussians.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
    gaussians.denom = torch.zeros((n, 1), device=dev)
    gaussians.max_radii2D = torch.zeros((n,), device=dev)
    return n


def reset_opacity_nonvisible(gaussians, visibility_filters) -> None:
    """`GaussianModel.reset_opacity_nonvisible(visibility_filters)` (gaussian_model.py:384-392; call site
    train_gaussians.py:260-263, every `gaussian_reset` iterations) on the reference's own model object, with the
    optimizer surgery of `replace_tensor_to_optimizer` (gaussian_model.py:477-490: the opacity group's Adam moments
    are zeroed, its step counter is kept).  Mirrors the reference exactly, including its quirk: a Gaussian seen by
    any of the filters keeps its ACTIVATED opacity as the new logit (`opacities_new[filter] = self.get_opacity[filter]`),
    every other one is reset to inverse_sigmoid(0.4).  A handful of elementwise device ops every ~2000 iterations:
    no kernel of its own, and no device->host synchronisation (the reference's boolean-mask stores each imply one)."""
    from torch import nn
    op = gaussians._opacity.detach()
    act = torch.sigmoid(op)                                   # get_opacity
    x = torch.ones_like(act) * 0.4
    new = torch.log(x / (1 - x))                              # inverse_sigmoid(0.4), general_utils.py:20-21
    seen = None
    for f in visibility_filters:
        f = f.reshape(-1).to(torch.bool)
        seen = f if seen is None else (seen | f)
    if seen is not None:
        new = torch.where(seen.view(-1, 1), act, new)
    opt = gaussians.optimizer
    for grp in opt.param_groups:
        if grp["name"] != "opacity":
            continue
        old = grp["params"][0]
        st = opt.state.pop(old, None)
        p = nn.Parameter(new.contiguous().requires_grad_(True))
        grp["params"][0] = p
        if st is not None:
            st["exp_avg"] = torch.zeros_like(p)
            st["exp_avg_sq"] = torch.zeros_like(p)
            opt.state[p] = st
        gaussians._opacity = p


def extend_from_pcd(gaussians, fused_point_cloud, features, scales, rots, opacities, markers, kp_scores) -> int:
    """Drop-in for `GaussianModel.extend_from_pcd(...)` (gaussian_model.py:222-241 -> densification_postfix :565-587 ->
    cat_tensors_to_optimizer :528-553; the key-frame insertion of train_gaussians.py:173-177) on the reference's own
    model object: the new rows are appended to the 8 parameter tensors and zero rows to the Adam moments of every group
    that has state in ONE launch (the reference: a torch.cat per tensor and per moment), new `nn.Parameter`s are bound
    to the model and to the optimizer's groups with the `step` counters carried over, and the densification
    statistics are reset, exactly as densification_postfix does.  `features` is [N, 3, (max_sh_degree + 1)^2]."""
    import ctypes as C
    from torch import nn
    lib = _native.load()
    dev = fused_point_cloud.device
    _require_gpu(fused_point_cloud, "fused_point_cloud")
    N = int(fused_point_cloud.shape[0])
    f32c = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()  # noqa: E731
    extra = {"xyz": f32c(fused_point_cloud), "f_dc": f32c(features[:, :, 0:1].transpose(1, 2)),
             "f_rest": f32c(features[:, :, 1:].transpose(1, 2)), "opacity": f32c(opacities), "marker": f32c(markers),
             "kp_score": f32c(kp_scores), "scaling": f32c(scales), "rotation": f32c(rots)}
    opt = gaussians.optimizer
    groups = {g["name"]: g for g in opt.param_groups}
    params = {k: getattr(gaussians, ATTR[k]).detach().contiguous() for k in GROUPS}
    P = int(params["xyz"].shape[0])
    widths = {k: _row_width(extra[k]) for k in GROUPS}
    for k in GROUPS:
        if P and _row_width(params[k]) != widths[k]:
            raise RuntimeError(f"extend_from_pcd: group `{k}` has rows of {_row_width(params[k])} floats, the new rows {widths[k]}")
        if int(extra[k].shape[0]) != N:
            raise RuntimeError(f"extend_from_pcd: `{k}` has {int(extra[k].shape[0])} rows, expected {N}")
    m, v = {}, {}
    for k in GROUPS:
        st = opt.state.get(groups[k]["params"][0], None)
        if st is not None and "exp_avg" in st:
            m[k], v[k] = st["exp_avg"].contiguous(), st["exp_avg_sq"].contiguous()
    n = P + N
    f32 = dict(dtype=torch.float32, device=dev)
    new_p = {k: torch.empty((n,) + tuple(extra[k].shape[1:]), **f32) for k in GROUPS}
    alloc = torch.empty_like if P else torch.zeros_like    # an empty model with state: nothing to copy, all rows new
    new_m = {k: alloc(new_p[k]) for k in m}
    new_v = {k: alloc(new_p[k]) for k in m}
    if n:
        src = _model_struct(params, P, widths)
        ext = _model_struct(extra, N, widths)
        out = _model_struct(new_p, n, widths)
        has = bool(m)
        m_in = _model_struct(m, P, widths) if has else None
        v_in = _model_struct(v, P, widths) if has else None
        m_out = _model_struct(new_m, n, widths) if has else None
        v_out = _model_struct(new_v, n, widths) if has else None
        ref = lambda s: None if s is None else C.byref(s)  # noqa: E731
        with _on_device(dev):
            _native.check(lib.splatraster_model_append(C.byref(src), ref(m_in), ref(v_in), C.byref(ext), C.byref(out),
                                                       ref(m_out), ref(v_out), _stream(dev)), "model_append")
    for k in GROUPS:
        grp = groups[k]
        old = grp["params"][0]
        st = opt.state.pop(old, None)
        p = nn.Parameter(new_p[k].requires_grad_(True))
        grp["params"][0] = p
        if st is not None:
            if k in new_m:
                st["exp_avg"], st["exp_avg_sq"] = new_m[k], new_v[k]
            opt.state[p] = st
        setattr(gaussians, ATTR[k], p)
    gaussians.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
    gaussians.denom = torch.zeros((n, 1), device=dev)
    gaussians.max_radii2D = torch.zeros((n,), device=dev)
    return n
