"""CPU oracle of SplatLoc's densify / clone / split / prune with optimizer-state surgery, of the Adam step
over the 8 parameter groups, and of the lr schedule — numpy restatement of the
reference's in-tree Python.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing on the product path).
Parity status: PINNED by tests/golden/densify.npz — parameters, Adam moments / step counters and
statistics recorded from the reference's own GaussianModel.densify_and_prune and torch.optim.Adam
(tests/golden/make_golden_densify.py; the split's torch.normal draw is injected as a recorded
standard-normal table indexed by source row).

  densify_and_prune       gaussian_model.py:655-675
    densify_and_clone     gaussian_model.py:632-653   rows appended in selection order
    densify_and_split     gaussian_model.py:590-630   N = 2 children per selected row, first all
                                                      "copy 0" children, then all "copy 1"; parents removed
    prune_points          gaussian_model.py:510-526   (+ _prune_optimizer :492-508)
    cat_tensors_to_optimizer / densification_postfix  gaussian_model.py:528-587 (zero moments for new
                                                      rows; statistics RESET to zero, incl. max_radii2D)
  Adam                    torch.optim.Adam(l, lr=0.0, eps=1e-15), gaussian_model.py:254-300
  lr schedule             general_utils.py:79-94 (helper), gaussian_model.py:311-325
"""
from __future__ import annotations

import numpy as np

GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "marker", "kp_score", "scaling", "rotation")
f32 = np.float32


def expon_lr(step, lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    if lr_delay_steps > 0:
        delay_rate = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
    else:
        delay_rate = 1.0
    t = np.clip(step / max_steps, 0, 1)
    return delay_rate * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)


def adam_step(params, grads, state, lrs, beta1=0.9, beta2=0.999, eps=1e-15):
    """One torch.optim.Adam step (no weight decay, no amsgrad).  params / grads / lrs: dict by group
    name; a group whose grad is None is skipped and gets no state (torch semantics).  state: dict
    name -> dict(m, v, step) (missing = not yet initialised).  float32 arithmetic like torch's."""
    params, state = dict(params), {k: dict(v) for k, v in state.items()}
    for name in GROUPS:
        g = grads.get(name)
        if g is None:
            continue
        p = np.asarray(params[name], f32)
        g = np.asarray(g, f32)
        st = state.get(name) or dict(m=np.zeros_like(p), v=np.zeros_like(p), step=0.0)
        step = st["step"] + 1.0
        m = (st["m"] + (g - st["m"]) * f32(1 - beta1)).astype(f32)           # exp_avg.lerp_(grad, 1 - beta1)
        v = (st["v"] * f32(beta2) + f32(1 - beta2) * g * g).astype(f32)
        bc1 = 1.0 - beta1 ** step
        bc2 = 1.0 - beta2 ** step
        step_size = lrs[name] / bc1
        denom = (np.sqrt(v) / f32(np.sqrt(bc2)) + f32(eps)).astype(f32)
        params[name] = (p - f32(step_size) * (m / denom)).astype(f32)
        state[name] = dict(m=m, v=v, step=step)
    return params, state


def _rotation_matrix(q):
    """general_utils.py:113-135 build_rotation (normalises the stored quaternion)."""
    q = np.asarray(q, f32)
    n = np.sqrt(q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1] + q[:, 2] * q[:, 2] + q[:, 3] * q[:, 3]).astype(f32)
    q = q / n[:, None]
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.zeros((q.shape[0], 3, 3), f32)
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - r * z)
    R[:, 0, 2] = 2 * (x * z + r * y)
    R[:, 1, 0] = 2 * (x * y + r * z)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - r * x)
    R[:, 2, 0] = 2 * (x * z - r * y)
    R[:, 2, 1] = 2 * (y * z + r * x)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def densify_and_prune(params, state, accum, denom, unit_noise, max_grad, min_opacity, extent, max_screen_size,
                      percent_dense, primitive_reg, N=2):
    """Returns (params', state', source_row, kind): the re-sized parameter groups and Adam state, and for
    every surviving row the source row it was made from and its kind (0 original, 1 clone, 2 / 3 the two
    split children).  Statistics after the call are all zero (densification_postfix), so they are not
    returned.  unit_noise [N, P, 3]: standard-normal table, row = source row of the split child.
    max_screen_size is accepted for signature parity: the postfix has zeroed max_radii2D before the final
    prune looks at it (gaussian_model.py:585-587, 664), so it can never trigger."""
    assert max_grad > 0.0, "clones are excluded from the split through their zero padded gradient"
    P = params["xyz"].shape[0]
    P0 = P
    with np.errstate(divide="ignore", invalid="ignore"):
        grads = (np.asarray(accum, f32) / np.asarray(denom, f32)).astype(f32)
    grads[np.isnan(grads)] = 0.0
    gn = np.abs(grads[:, 0])                                   # torch.norm over a 1-wide last dim
    scal = np.exp(np.asarray(params["scaling"], f32)).astype(f32)
    smax = scal.max(axis=1)
    thr_sz = f32(percent_dense * extent)
    # ---- clone ----
    clone = (gn >= f32(max_grad)) & (smax <= thr_sz)
    cur = {k: np.concatenate((np.asarray(params[k], f32), np.asarray(params[k], f32)[clone]), 0) for k in GROUPS}
    src = np.concatenate((np.arange(P), np.nonzero(clone)[0]))
    kind = np.concatenate((np.zeros(P, np.int64), np.ones(int(clone.sum()), np.int64)))
    mom = {}
    for k, st in state.items():
        z = np.zeros_like(np.asarray(params[k], f32)[clone])
        mom[k] = dict(m=np.concatenate((st["m"], z), 0), v=np.concatenate((st["v"], z), 0), step=st["step"])
    P1 = cur["xyz"].shape[0]
    # ---- split ----
    padded = np.zeros(P1, f32)
    padded[:P] = grads[:, 0]
    scal1 = np.exp(cur["scaling"]).astype(f32)
    split = (padded >= f32(max_grad)) & (scal1.max(axis=1) > thr_sz)
    assert not split[P:].any()
    sel = np.nonzero(split)[0]
    stds = np.tile(scal1[split], (N, 1))
    noise = np.concatenate([np.asarray(unit_noise, f32)[c, sel] for c in range(N)], 0)
    samples = (noise * stds).astype(f32)
    R = np.tile(_rotation_matrix(cur["rotation"][split]), (N, 1, 1))
    new_xyz = (np.einsum("nij,nj->ni", R, samples).astype(f32) + np.tile(cur["xyz"][split], (N, 1))).astype(f32)
    new = {k: np.tile(cur[k][split], (N,) + (1,) * (cur[k].ndim - 1)) for k in GROUPS}
    new["xyz"] = new_xyz
    new["scaling"] = np.log(np.tile(scal1[split], (N, 1)) / f32(0.8 * N)).astype(f32)
    ns = int(split.sum())
    keep = np.concatenate((~split, np.ones(N * ns, bool)))
    cur = {k: np.concatenate((cur[k], new[k]), 0)[keep] for k in GROUPS}
    src = np.concatenate((src, np.tile(sel, N)))[keep]
    kind = np.concatenate((kind, np.repeat(np.arange(N) + 2, ns)))[keep]
    for k in mom:
        z = np.zeros_like(new[k])
        mom[k] = dict(m=np.concatenate((mom[k]["m"], z), 0)[keep], v=np.concatenate((mom[k]["v"], z), 0)[keep],
                      step=mom[k]["step"])
    # ---- final prune ----
    opac = (1.0 / (1.0 + np.exp(-cur["opacity"][:, 0]))).astype(f32)
    prune = opac < f32(min_opacity)
    if max_screen_size:
        big_ws = np.exp(cur["scaling"]).astype(f32).max(axis=1) > f32(0.1 * extent)
        prune = prune | big_ws           # big_points_vs: max_radii2D is all zero here
    if primitive_reg:
        prune = prune & (cur["marker"][:, 0] <= f32(0.005))
    keep = ~prune
    cur = {k: cur[k][keep] for k in GROUPS}
    for k in mom:
        mom[k] = dict(m=mom[k]["m"][keep], v=mom[k]["v"][keep], step=mom[k]["step"])
    assert P0 == P
    return cur, mom, src[keep], kind[keep]
