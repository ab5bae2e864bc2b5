"""The two inner loops of SplatLoc as the rasterizer sees them, with the device-side pieces of this package in place
of the reference's chains of torch ops: drop-in bodies for `SplatLoc.color_refinement`'s iteration
(train_gaussians.py:272-297 — 26 000 of the ~35 000 rasterizer calls of a scene) and the learning-rate schedule it
calls.  `gaussians` is the reference's own GaussianModel object (or anything with the same attributes); its optimizer
may be torch.optim.Adam or splatloc_amd.optim.Adam (one launch over the 8 groups, key-primitive gate folded in).
"""
from __future__ import annotations

import os

import math

import torch

from .densify import add_densification_stats_window
from .fused import render_window
from .losses import refinement_loss_and_grad


def expon_lr(step, lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000) -> float:
    """`helper` of gaussian_splatting/utils/general_utils.py:79-94 (the xyz learning-rate schedule)."""
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    if lr_delay_steps > 0:
        delay_rate = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0.0), 1.0))
    else:
        delay_rate = 1.0
    t = min(max(step / max_steps, 0.0), 1.0)
    return delay_rate * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)


def update_learning_rate(gaussians, iteration) -> float:
    """GaussianModel.update_learning_rate (gaussian_model.py:311-326): the model's own method when it has one."""
    if hasattr(gaussians, "update_learning_rate"):
        return gaussians.update_learning_rate(iteration)
    for grp in gaussians.optimizer.param_groups:
        if grp["name"] == "xyz":
            grp["lr"] = expon_lr(iteration, lr_init=gaussians.lr_init, lr_final=gaussians.lr_final,
                                 lr_delay_mult=gaussians.lr_delay_mult, max_steps=gaussians.max_steps)
            return grp["lr"]
    return 0.0


def _direct_refine_ok(gaussians, pipe) -> bool:
    """SplatLoc's own configuration (SH degree 0, colours converted in Python, covariance in the rasterizer) on a non-empty
    model whose tensors all take gradients: the iteration can run without autograd (below)."""
    if not bool(pipe.convert_SHs_python) or bool(pipe.compute_cov3D_python) or int(gaussians.active_sh_degree) != 0:
        return False
    if int(gaussians._xyz.shape[0]) == 0 or not gaussians._xyz.is_cuda:
        return False
    return all(getattr(gaussians, a).requires_grad for a in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_kp_score",
                                                             "_scaling", "_rotation"))


RAW_BACKWARD = os.environ.get("SPLATLOC_RAW_BACKWARD", "1") != "0"     # (A/B and tests: 0 = the two-kernel chain)


def _raw_backward_ok(gaussians) -> bool:
    """The raw-parameter kernels cover SplatLoc's own layout: contiguous fp32 [P,3] log-scales, [P,4] quaternions, [P,1] logits,
    [P,1,3] SH dc, no higher SH coefficients, ONE key-point column (C = 4)."""
    if not RAW_BACKWARD:
        return False
    g = gaussians
    ok = lambda t, shp: (t.dtype is torch.float32 and t.is_contiguous() and tuple(t.shape[1:]) == shp and not (t.data_ptr() & 15))  # noqa: E731
    return (ok(g._scaling, (3,)) and ok(g._rotation, (4,)) and ok(g._opacity, (1,)) and ok(g._features_dc, (1, 3))
            and int(g._features_rest.shape[1]) == 0 and g._kp_score.dim() == 2 and ok(g._kp_score, (int(g._kp_score.shape[1]),))
            and int(g._kp_score.shape[1]) == 1)     # (C = 4: the accumulator rows' colour columns share the moments' 64-byte line)


def _color_refinement_step_direct(viewpoint_cam, gaussians, background, lambda_dssim, iteration, primitive_reg):
    """The iteration of `color_refinement_step` with the SAME forward / backward code (`_ActivatePack`, `_RasterizeWindow`:
    their static forward / backward called directly with a plain context) but without building an autograd graph: the
    refinement iteration is HOST-bound at SplatLoc's frame size (tools/refine_idle.py: 0.42 ms of Python per iteration against
    0.25 - 0.55 ms of kernels), and a third of that host time was the autograd engine handing the two backward nodes to its
    worker thread.  Gradients land in `.grad` exactly as autograd would leave them: `_xyz` <- dL/dmeans3D, `_features_rest`
    <- its empty gradient, `_kp_score` <- the zero column, `_marker` <- nothing (tests/test_gpu_refine.py, both paths)."""
    from .fused import _ActivatePack, _view_settings
    from .rasterizer import PlainCtx, _RasterizeWindow
    with torch.no_grad():
        xyz = gaussians._xyz
        settings = _view_settings(viewpoint_cam, gaussians, background, 1.0)
        c_ras = PlainCtx()
        raw_ok = _raw_backward_ok(gaussians) and int(xyz.shape[0]) > 0
        if raw_ok:
            # the activations run inside the projection kernel (rasterizer.py: ctx.raw_fwd): it fills these four tensors
            P = int(xyz.shape[0])
            f32 = dict(dtype=torch.float32, device=xyz.device)
            scales, rotations, opacity = torch.empty((P, 3), **f32), torch.empty((P, 4), **f32), torch.empty((P, 1), **f32)
            colors = torch.empty((P, 3 + int(gaussians._kp_score.shape[1])), **f32)
            c_ras.raw_fwd = (gaussians._scaling.detach(), gaussians._rotation.detach(), gaussians._opacity.detach(),
                             gaussians._features_dc.detach(), gaussians._kp_score.detach())
            c_act = None
        else:
            c_act = PlainCtx()
            scales, rotations, opacity, colors = _ActivatePack.forward(
                c_act, xyz, gaussians._features_dc, gaussians._features_rest, gaussians._scaling, gaussians._rotation,
                gaussians._opacity, gaussians._kp_score, None, 0)
        rgb, _kp, _depth, _alpha, radii = _RasterizeWindow.forward(c_ras, xyz, colors, opacity, scales, rotations, None, (settings,),
                                                                 3, None, xyz)     # (means2D is a gradient carrier only)
        gt_image = viewpoint_cam.original_image
        if gt_image.device != rgb.device:
            gt_image = gt_image.to(rgb.device)
        loss, g_image = refinement_loss_and_grad(rgb, gt_image, lambda_dssim)
        if raw_ok:
            # the rasterizer's backward writes the RAW parameters' gradients itself (rasterizer.py: ctx.raw): two launches and the
            # activated-gradient tensors less per iteration, bit-identical values (tests/test_gpu_refine.py)
            c_ras.raw = (gaussians._scaling.detach(), gaussians._rotation.detach(), gaussians._opacity.detach(),
                         gaussians._features_dc.detach(), gaussians._kp_score.detach())
            d = _RasterizeWindow.backward(c_ras, g_image, None, None, None, None)
            d_m3 = d[0]
            d_sc, d_ro, d_opa, d_fd, d_ex = c_ras.raw_out
            d_fr = torch.empty_like(gaussians._features_rest)
        else:
            d = _RasterizeWindow.backward(c_ras, g_image, None, None, None, None)
            d_m3, d_col, d_op, d_sca, d_rot = d[0], d[1], d[2], d[3], d[4]
            _dx, d_fd, d_fr, d_sc, d_ro, d_opa, d_ex, _, _ = _ActivatePack.backward(c_act, d_sca, d_rot, d_op, d_col)
        for p, g in ((gaussians._xyz, d_m3), (gaussians._features_dc, d_fd), (gaussians._features_rest, d_fr),
                     (gaussians._scaling, d_sc), (gaussians._rotation, d_ro), (gaussians._opacity, d_opa), (gaussians._kp_score, d_ex)):
            if g is not None:
                p.grad = g if p.grad is None else p.grad + g
        opt = gaussians.optimizer
        if primitive_reg:
            if hasattr(opt, "set_key_gate"):
                opt.set_key_gate(gaussians._marker, 0.005)
            else:
                key_mask = gaussians._marker.detach().squeeze() > 0.005
                gaussians._xyz.grad[key_mask] = 0
        elif hasattr(opt, "set_key_gate"):
            opt.set_key_gate(None)
        if hasattr(opt, "set_radii_update") and radii.dtype == torch.int32 and radii.is_contiguous():
            opt.set_radii_update(radii, gaussians.max_radii2D)      # the statistics line rides the fused Adam launch
        else:
            add_densification_stats_window(None, [radii], None, None, gaussians.max_radii2D)
        opt.step()
        opt.zero_grad(set_to_none=True)
        update_learning_rate(gaussians, iteration)
    return loss


def color_refinement_step(viewpoint_cam, gaussians, pipe, background, lambda_dssim: float, iteration: int,
                          primitive_reg: bool = True, render_path: str = "auto"):
    """One iteration of SplatLoc.color_refinement (train_gaussians.py:275-297):

        render -> (1 - l) L1 + l (1 - SSIM) on the RGB channels -> backward -> key-primitive gate on xyz.grad ->
        max_radii2D update -> optimizer.step -> zero_grad -> update_learning_rate(iteration)

    with: ONE window-of-one launch sequence whose `render` / `kp_prob` / depth / opacity are separate autograd outputs —
    only `render` reaches the loss, so the backward kernel runs on 3 colour channels without the depth / alpha terms
    and no zero-padded gradient images are built; the fused L1 + SSIM loss (two kernels instead of five grouped 11x11
    convolutions and their autograd backward); the `max_radii2D` line as one launch without boolean-mask indexing
    (the reference: two `nonzero` + a device->host sync); the gate inside the fused Adam launch when the optimizer is
    splatloc_amd.optim.Adam (else the reference's masked assignment).  Returns the loss tensor (no host sync).
    `render_path`: "auto" (graph-free when the configuration allows it), "window" (the window-of-one under autograd) or
    "per-view" (the drop-in `render()` -> `diff_gauss.GaussianRasterizer` call an unmodified train_gaussians.py issues)."""
    if render_path not in ("auto", "window", "per-view"):
        raise ValueError(f"color_refinement_step: unknown render_path {render_path!r}")
    if render_path == "auto" and _direct_refine_ok(gaussians, pipe):
        return _color_refinement_step_direct(viewpoint_cam, gaussians, background, lambda_dssim, iteration, primitive_reg)
    if render_path == "per-view":
        from .fused import render
        pkg = render(viewpoint_cam, gaussians, pipe, background)
    else:
        pkg = render_window([viewpoint_cam], gaussians, pipe, background, batched=True)[0][0]
    if pkg is None:
        return None
    image, radii = pkg["render"], pkg["radii"]
    gt_image = viewpoint_cam.original_image.to(image.device)
    loss, g_image = refinement_loss_and_grad(image, gt_image, lambda_dssim)   # the fused launch holds the gradient: no loss node
    image.backward(g_image)
    opt = gaussians.optimizer
    with torch.no_grad():
        if primitive_reg:
            if hasattr(opt, "set_key_gate"):
                opt.set_key_gate(gaussians._marker, 0.005)
            else:
                key_mask = gaussians._marker.detach().squeeze() > 0.005
                gaussians._xyz.grad[key_mask] = 0
        elif hasattr(opt, "set_key_gate"):
            opt.set_key_gate(None)
        add_densification_stats_window(None, [radii], None, None, gaussians.max_radii2D)
        opt.step()
        opt.zero_grad(set_to_none=True)
        update_learning_rate(gaussians, iteration)
    return loss


def _map_grads_direct(mine, gaussians, pipe, background, config, with_reg: bool):
    """The gradient part of `map_step` without an autograd graph (same forward / backward code as the graph path: the static
    `forward` / `backward` of `_ActivatePack`, `_RasterizeWindow`, `_IsotropicLoss` with a plain context; the per-view losses
    already carry their gradients).  At SplatLoc's frame size a map step is ~2 ms of kernels under ~3 ms of Python (tools/
    hostprof_steps.py): the engine's hand-off of three backward nodes and its validation of 16 explicit gradient tensors were a
    third of it.  Leaves the raw-parameter gradients in `.grad`.  Returns (pkgs, loss, [dL/dmeans2D per view]) or None when the
    configuration needs the general path (view-dependent colours, python covariance, mixed image sizes, an empty model)."""
    from .fused import _ActivatePack, _view_settings
    from .losses import _IsotropicLoss, mapping_loss_window
    from .rasterizer import PlainCtx, _RasterizeWindow, _window_compatible
    from . import _native
    if not _direct_refine_ok(gaussians, pipe) or len(mine) > _native.MAX_WINDOW_VIEWS:
        return None
    P = int(gaussians._xyz.shape[0])
    if P * len(mine) > (1 << 24):          # (rasterize_window would chunk the window: general path)
        return None
    settings = [_view_settings(vp, gaussians, background, 1.0) for vp in mine]
    if not _window_compatible(settings):
        return None
    with torch.no_grad():
        xyz = gaussians._xyz
        c_ras = PlainCtx()
        V = len(mine)
        raw_ok = _raw_backward_ok(gaussians) and P > 0
        raw_in = (gaussians._scaling.detach(), gaussians._rotation.detach(), gaussians._opacity.detach(),
                  gaussians._features_dc.detach(), gaussians._kp_score.detach()) if raw_ok else None
        if raw_ok:
            # the activations run inside the projection kernel (rasterizer.py: ctx.raw_fwd), which fills these four tensors
            f32 = dict(dtype=torch.float32, device=xyz.device)
            scales, rotations, opacity = torch.empty((P, 3), **f32), torch.empty((P, 4), **f32), torch.empty((P, 1), **f32)
            colors = torch.empty((P, 3 + int(gaussians._kp_score.shape[1])), **f32)
            c_ras.raw_fwd = raw_in
            c_act = None
        else:
            c_act = PlainCtx()
            scales, rotations, opacity, colors = _ActivatePack.forward(
                c_act, xyz, gaussians._features_dc, gaussians._features_rest, gaussians._scaling, gaussians._rotation,
                gaussians._opacity, gaussians._kp_score, None, 0)
        outs = _RasterizeWindow.forward(c_ras, xyz, colors, opacity, scales, rotations, None, tuple(settings), 3, None,
                                        *([xyz] * V))
        pkgs = []
        for v in range(V):
            rgb, kp, depth, alpha, radii = outs[5 * v:5 * v + 5]
            pkgs.append({"render": rgb, "kp_prob": kp, "depth": depth, "opacity": alpha, "radii": radii})
        _t, g, loss = mapping_loss_window(config, pkgs, mine)      # g = [g_render, g_depth, g_kp] per view
        gouts = []
        for v in range(V):
            gouts += [g[3 * v], g[3 * v + 2], g[3 * v + 1], None, None]      # (rgb, last, depth, alpha, radii)
        reg = None
        if with_reg:
            # 0.01 * isotropic regulariser on exp(_scaling) (train_gaussians.py:221-228): its gradient w.r.t. the ACTIVATED scales
            # joins the rasterizer's before the activation backward multiplies by exp(s)
            c_reg = PlainCtx()
            value = _IsotropicLoss.forward(c_reg, scales, gaussians._marker.detach())
            row_grad, out = c_reg.saved_tensors
            reg = (row_grad, out, 0.01)
            loss = 0.01 * value if loss is None else loss + 0.01 * value
        if raw_ok:
            # the rasterizer's backward writes the RAW parameters' gradients itself, the regulariser's term included (ctx.raw)
            c_ras.raw = raw_in + ((reg,) if reg is not None else ())
            d = _RasterizeWindow.backward(c_ras, *gouts)
            d_m3 = d[0]
            grads2d = list(d[9:9 + V])
            d_sc, d_ro, d_opa, d_fd, d_ex = c_ras.raw_out
            d_fr = torch.empty_like(gaussians._features_rest)
        else:
            d = _RasterizeWindow.backward(c_ras, *gouts)
            d_m3, d_col, d_op, d_sca, d_rot = d[0], d[1], d[2], d[3], d[4]
            grads2d = list(d[9:9 + V])
            if reg is not None:
                d_sca = d_sca + ((reg[2] * reg[1][1]) * reg[0]).view(-1, 1)
            _dx, d_fd, d_fr, d_sc, d_ro, d_opa, d_ex, _, _ = _ActivatePack.backward(c_act, d_sca, d_rot, d_op, d_col)
        for p, gr in ((gaussians._xyz, d_m3), (gaussians._features_dc, d_fd), (gaussians._features_rest, d_fr),
                      (gaussians._scaling, d_sc), (gaussians._rotation, d_ro), (gaussians._opacity, d_opa), (gaussians._kp_score, d_ex)):
            if gr is not None:
                p.grad = gr if p.grad is None else p.grad + gr
    return pkgs, loss, grads2d


# how a multi-GPU map_step sums its payload: "ring" (one all-reduce) or "rs_ag" (reduce-scatter + all-gather);
# frame_parallel.reduce_step(mode=...).  bench.py --reduce sets it; every rank must use the same value
REDUCE_MODE = os.environ.get("SPLATLOC_REDUCE_MODE", "ring")
# what the last multi-GPU map_step exchanged (frame_parallel.reduce_step's info: collectives, path, bytes) — monitoring
LAST_STEP_INFO: dict = {}


def map_step(viewpoints, gaussians, pipe, background, config, iteration_count: int, *, densify=None,
             gaussian_reset: int = 0, seed: int = 0, group=None, render_path: str = "auto", distributed: bool = True):
    """One iteration of the loop body of SplatLoc.map (train_gaussians.py:188-267) on the window `viewpoints` (the
    caller has drawn it: `all_viewpoint_stack[torch.randperm(len(...))[:window_size]]`, :195), with the device-side
    pieces of this package, single- or multi-GPU:

        render the window + per-view get_loss_mapping + get_loss_marker      -> fused.render_window (ONE launch sequence)
        + 0.01 * isotropic regulariser (primitive_reg)                        -> losses.isotropic_loss (no .cpu() mask)
        backward                                                              -> one window backward (gradients summed in-kernel)
        key-primitive gate, max_radii2D / add_densification_stats per view    -> one statistics launch
        densify_and_prune every `densify["every"]` iterations (offset)        -> densify.densify_and_prune
        reset_opacity_nonvisible every `gaussian_reset` iterations            -> densify.reset_opacity_nonvisible
        optimizer.step, zero_grad, update_learning_rate(iteration_count)

    Frame-parallel data parallelism (torch.distributed initialised, one process per GPU, a full replica of
    the scene per rank): the views of the window are dealt round-robin to the ranks (frame_parallel.shard_views); ONE SUM
    all-reduce carries the parameter gradients and the statistics increments, ONE MAX all-reduce max_radii2D and (on a
    reset step) the visibility union — two collectives per step (frame_parallel.reduce_step; a rank without views, world
    size > window size, contributes zeros) —, so every replica takes the SAME optimizer step and densifies identically (the split noise is
    counter-based: keyed by (seed, iteration_count, source row, copy)) — the replicas stay bit-identical without ever
    broadcasting parameters.  The regulariser is added on rank 0 only (the reduced gradient contains it once).
    `densify`: dict(grad_threshold, min_opacity, extent, size_threshold, every, offset) or None.
    `render_path`: "auto" (the graph-free window path when the configuration allows it, else the window path under autograd),
    "window" (always under autograd) or "per-view" (the reference's loop of per-view render() calls through the drop-in
    autograd.Function — what an unmodified train_gaussians.py issues); LAST_STEP_INFO["render_path"] says which one ran.
    Returns the rank's loss tensor (None on a rank without work)."""
    import torch.distributed as dist
    from .densify import densify_and_prune, reset_opacity_nonvisible
    from .frame_parallel import collectives_active, reduce_step, shard_views
    from .losses import isotropic_loss, mapping_loss_window
    # `distributed=False`: this rank reconstructs its OWN scene although a process group exists (one scene per GPU,
    # /root/reference/replica.sh; bench.py --stage scene --replicas): no collective at all.  (A group of ONE rank exchanges
    # nothing either, unless SPLATLOC_FORCE_COLLECTIVES=1 asks for the collectives anyway: the one-GPU RCCL contact test.)
    multi = distributed and collectives_active(group)
    rank = dist.get_rank(group) if multi else 0
    world = dist.get_world_size(group) if multi else 1
    primitive_reg = bool(config["Training"].get("primitive_reg", True))
    viewpoints = list(viewpoints)
    mine = [viewpoints[i] for i in shard_views(list(range(len(viewpoints))), rank, world)]
    if render_path not in ("auto", "window", "per-view"):
        raise ValueError(f"map_step: unknown render_path {render_path!r}")
    direct = None
    if mine and render_path == "auto":
        direct = _map_grads_direct(mine, gaussians, pipe, background, config, primitive_reg and rank == 0)
    ran = "direct-window" if direct is not None else ("per-view" if render_path == "per-view" else "window")
    if direct is not None:
        pkgs, loss, grads2d_direct = direct
    else:
        grads2d_direct = None
        per_view = render_path == "per-view"        # the literal loop: one render() per view, each with its own activations
        pkgs, _ = render_window(mine, gaussians, pipe, background, batched=not per_view, share_activations=not per_view)
        pairs = [(p, v) for p, v in zip(pkgs, mine) if p is not None]      # views and packages filtered TOGETHER
        pkgs, mine = [p for p, _ in pairs], [v for _, v in pairs]
        # the per-view losses carry their own gradients (one fused launch each): ONE backward on the rasterizer's outputs, no
        # per-view loss nodes / gradient scalings / additions (losses.mapping_loss_window)
        tensors, grads, loss = mapping_loss_window(config, pkgs, mine)
        if primitive_reg and rank == 0 and gaussians._xyz.shape[0] > 0:
            reg = 0.01 * isotropic_loss(torch.exp(gaussians._scaling), gaussians._marker)
            tensors, grads = tensors + [reg], grads + [None]
            loss = reg.detach() if loss is None else loss + reg.detach()
        if tensors:
            torch.autograd.backward(tensors, grads)
    params = [getattr(gaussians, a) for a in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_kp_score", "_scaling",
                                              "_rotation")]      # `_marker` never receives a gradient in map()
    opt = gaussians.optimizer
    with torch.no_grad():
        P = int(gaussians._xyz.shape[0])
        dev = gaussians._xyz.device
        grads2d = grads2d_direct if grads2d_direct is not None else [p["viewspace_points"].grad for p in pkgs]
        radii = [p["radii"] for p in pkgs]
        update_gaussian = bool(densify) and iteration_count % int(densify["every"]) == int(densify.get("offset", 0))
        reset_now = bool(gaussian_reset) and iteration_count % gaussian_reset == 0 and not update_gaussian
        seen = None
        if reset_now:       # the union of the window's visibility masks (gaussian_model.py:384-392)
            seen = torch.zeros(P, dtype=torch.float32, device=dev)
            for p in pkgs:
                seen = torch.maximum(seen, (p["radii"] > 0).to(torch.float32))      # visibility_filter = radii > 0
        if multi:
            # everything the replicas exchange in this step, in TWO collectives (frame_parallel.reduce_step):
            #   SUM over [parameter gradients | increments of xyz_gradient_accum, denom]
            #   MAX over [max_radii2D | visibility flags of a reset step]
            for p in params:        # a rank without views: zero gradients (incl. the empty `_features_rest` group, so that
                if p.grad is None:  # every replica's optimizer creates the same state)
                    p.grad = torch.zeros_like(p)
            live = [p for p in params if p.numel()]
            inc = torch.zeros((2, P, 1), device=dev)
            if pkgs:
                add_densification_stats_window(grads2d, radii, inc[0], inc[1], gaussians.max_radii2D)
            g_out, inc_out, info = reduce_step([p.grad for p in live], sum_extras=[inc[0], inc[1]],
                                               max_extras=[gaussians.max_radii2D] + ([seen] if seen is not None else []),
                                               group=group, mode=REDUCE_MODE, force=True)
            for p, g in zip(live, g_out):
                p.grad = g          # views of the reduced buffer: no copy back
            gaussians.xyz_gradient_accum += inc_out[0]
            gaussians.denom += inc_out[1]
            LAST_STEP_INFO.clear()
            LAST_STEP_INFO.update(info)
        else:
            LAST_STEP_INFO.clear()
            if pkgs:
                add_densification_stats_window(grads2d, radii, gaussians.xyz_gradient_accum, gaussians.denom, gaussians.max_radii2D)
        LAST_STEP_INFO["render_path"] = ran
        if primitive_reg:
            if hasattr(opt, "set_key_gate"):
                opt.set_key_gate(gaussians._marker, 0.005)
            elif gaussians._xyz.grad is not None:
                gaussians._xyz.grad[gaussians._marker.detach().squeeze() > 0.005] = 0
        elif hasattr(opt, "set_key_gate"):
            opt.set_key_gate(None)
        if update_gaussian:
            densify_and_prune(gaussians, densify["grad_threshold"], densify["min_opacity"], densify["extent"],
                              densify["size_threshold"], seed=seed, draw_id=iteration_count)
        if reset_now:
            reset_opacity_nonvisible(gaussians, [seen > 0])
        opt.step()
        opt.zero_grad(set_to_none=True)
        update_learning_rate(gaussians, iteration_count)
    return loss
