"""The two inner loops of SplatLoc as the rasterizer sees them, with the device-side pieces of this package in place
of the reference's chains of torch ops: drop-in bodies for `SplatLoc.color_refinement`'s iteration
(train_gaussians.py:272-297 — 26 000 of the ~35 000 rasterizer calls of a scene) and the learning-rate schedule it
calls.  `gaussians` is the reference's own GaussianModel object (or anything with the same attributes); its optimizer
may be torch.optim.Adam or splatloc_amd.optim.Adam (one launch over the 8 groups, key-primitive gate folded in).
"""
from __future__ import annotations

import math

import torch

from .densify import add_densification_stats_window
from .fused import render_window
from .losses import refinement_loss


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


def color_refinement_step(viewpoint_cam, gaussians, pipe, background, lambda_dssim: float, iteration: int,
                          primitive_reg: bool = True):
    """One iteration of SplatLoc.color_refinement (train_gaussians.py:275-297):

        render -> (1 - l) L1 + l (1 - SSIM) on the RGB channels -> backward -> key-primitive gate on xyz.grad ->
        max_radii2D update -> optimizer.step -> zero_grad -> update_learning_rate(iteration)

    with: ONE window-of-one launch sequence whose `render` / `kp_prob` / depth / opacity are separate autograd outputs —
    only `render` reaches the loss, so the backward kernel runs on 3 colour channels without the depth / alpha terms
    and no zero-padded gradient images are built; the fused L1 + SSIM loss (two kernels instead of five grouped 11x11
    convolutions and their autograd backward); the `max_radii2D` line as one launch without boolean-mask indexing
    (the reference: two `nonzero` + a device->host sync); the gate inside the fused Adam launch when the optimizer is
    splatloc_amd.optim.Adam (else the reference's masked assignment).  Returns the loss tensor (no host sync)."""
    pkgs, _ = render_window([viewpoint_cam], gaussians, pipe, background, batched=True)
    pkg = pkgs[0]
    if pkg is None:
        return None
    image, radii = pkg["render"], pkg["radii"]
    gt_image = viewpoint_cam.original_image.to(image.device)
    loss = refinement_loss(image, gt_image, lambda_dssim)
    loss.backward()
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
