"""oracle/densify.py (numpy) against tests/golden/densify.npz — values recorded from the reference's own
GaussianModel.densify_and_prune, torch.optim.Adam over its 8 groups, and the lr helper
(tests/golden/make_golden_densify.py)."""
import os

import numpy as np
import pytest

from oracle import densify as od

G = od.GROUPS


def _load(golden_dir):
    return np.load(os.path.join(golden_dir, "densify.npz"))


def _params(d, pre):
    return {k: d[pre + k] for k in G}


def _state(d, pre):
    return {k: dict(m=d[f"{pre}m_{k}"], v=d[f"{pre}v_{k}"], step=float(d[f"{pre}step_{k}"])) for k in G
            if f"{pre}m_{k}" in d.files}


def _lrs(d, pre):
    return {k: float(d[f"{pre}lr_{k}"]) for k in G}


def _assert_params(got, d, pre, rtol, atol):
    for k in G:
        assert got[k].shape == d[pre + k].shape, (k, got[k].shape, d[pre + k].shape)
        np.testing.assert_allclose(got[k], d[pre + k], rtol=rtol, atol=atol, err_msg=k)


def run_adam(d, case, leg, n, params, state, first_iteration):
    """`n` steps with the recorded gradients; lr of xyz follows update_learning_rate AFTER each step."""
    lrs = _lrs(d, f"{case}{'s0_' if leg == 'a_' else 's2_'}")
    for k in range(n):
        grads = {g: (d[f"{case}{leg}grad{k}_{g}"] if f"{case}{leg}grad{k}_{g}" in d.files else None) for g in G}
        params, state = od.adam_step(params, grads, state, lrs)
        lrs["xyz"] = od.expon_lr(first_iteration + k, lr_init=0.0016 * 6.0, lr_final=0.0000016 * 6.0,
                                 lr_delay_mult=0.01, max_steps=30000)
        np.testing.assert_allclose(lrs["xyz"], float(d[f"{case}{leg}xyz_lr_after{k}"]), rtol=1e-12)
    return params, state


def test_lr_schedule(golden_dir):
    d = _load(golden_dir)
    got = [od.expon_lr(int(s), lr_init=0.0016 * 6.0, lr_final=0.0000016 * 6.0, lr_delay_mult=0.01, max_steps=30000)
           for s in d["lr_steps"]]
    np.testing.assert_allclose(got, d["lr_values"], rtol=1e-12)
    got = [od.expon_lr(int(s), lr_init=1e-2, lr_final=1e-4, lr_delay_steps=500, lr_delay_mult=0.01, max_steps=30000)
           for s in d["lr_steps"]]
    np.testing.assert_allclose(got, d["lr_values_delay"], rtol=1e-12)


@pytest.mark.parametrize("case", ["plain_", "reg_"])
def test_adam_and_densify_against_reference(golden_dir, case):
    d = _load(golden_dir)
    extent, pd, max_grad, min_opacity, size_thr = (float(v) for v in d[case + "hyper"])
    # ---- 3 Adam steps from the initial state (no moments yet; the marker never gets a gradient) ----
    params, state = run_adam(d, case, "a_", 3, _params(d, case + "s0_"), {}, 1)
    assert "marker" not in state and f"{case}s1_m_marker" not in d.files            # _marker.grad is None
    assert d[case + "s0_f_rest"].shape[1:] == (0, 3)                                 # f_rest is [P, 0, 3]
    _assert_params(params, d, case + "s1_", rtol=2e-6, atol=1e-7)
    ref_state = _state(d, case + "s1_")
    for k in ref_state:
        np.testing.assert_allclose(state[k]["m"], ref_state[k]["m"], rtol=1e-5, atol=1e-10)
        np.testing.assert_allclose(state[k]["v"], ref_state[k]["v"], rtol=1e-5, atol=1e-14)
        assert state[k]["step"] == ref_state[k]["step"] == 3.0
    # ---- densify_and_prune from the REFERENCE's state (so errors do not compound) ----
    params, state = _params(d, case + "s1_"), ref_state
    new_p, new_s, src, kind = od.densify_and_prune(params, state, d[case + "accum_in"], d[case + "denom_in"],
                                                   d[case + "unit_noise"], max_grad, min_opacity, extent, size_thr, pd,
                                                   primitive_reg=(case == "reg_"))
    P0, P1 = params["xyz"].shape[0], d[case + "s2_xyz"].shape[0]
    assert new_p["xyz"].shape[0] == P1 and P1 != P0
    assert (kind == 1).any() and (kind == 2).any() and (kind == 3).any() and (kind == 0).sum() < P0
    for k in G:   # copies are exact; the split's xyz / scaling go through a rotation / log
        tol = dict(rtol=2e-6, atol=2e-6) if k in ("xyz", "scaling") else dict(rtol=0, atol=0)
        np.testing.assert_allclose(new_p[k], d[case + "s2_" + k], err_msg=k, **tol)
    ref2 = _state(d, case + "s2_")
    assert set(ref2) == set(new_s)
    for k in ref2:
        assert np.array_equal(new_s[k]["m"], ref2[k]["m"]) and np.array_equal(new_s[k]["v"], ref2[k]["v"]), k
        assert new_s[k]["step"] == ref2[k]["step"] == 3.0                            # the step counter carries over
        assert not new_s[k]["m"][kind != 0].any()                                    # new rows start from zero moments
    assert not d[case + "s2_accum"].any() and not d[case + "s2_denom"].any() and not d[case + "s2_max_radii"].any()
    assert d[case + "s2_accum"].shape == (P1, 1) and d[case + "s2_max_radii"].shape == (P1,)
    # ---- 2 more steps on the re-sized model ----
    params, state = run_adam(d, case, "b_", 2, _params(d, case + "s2_"), ref2, 4)
    _assert_params(params, d, case + "s3_", rtol=2e-6, atol=1e-7)
    for k, st in _state(d, case + "s3_").items():
        assert state[k]["step"] == st["step"] == 5.0
        np.testing.assert_allclose(state[k]["m"], st["m"], rtol=1e-5, atol=1e-10)


def test_reset_opacity_nonvisible_matches_reference(golden_dir):
    """splatloc_amd.densify.reset_opacity_nonvisible (elementwise torch ops, runs on any device) against the state
    recorded from the reference's GaussianModel.reset_opacity_nonvisible + replace_tensor_to_optimizer."""
    import types
    import torch
    from splatloc_amd.densify import reset_opacity_nonvisible
    d = np.load(os.path.join(golden_dir, "reset_opacity.npz"))
    p = torch.nn.Parameter(torch.from_numpy(d["opacity_before"]).requires_grad_(True))
    other = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.Adam([{"params": [other], "lr": 0.1, "name": "xyz"}, {"params": [p], "lr": 0.05, "name": "opacity"}], lr=0.0, eps=1e-15)
    opt.state[p] = {"step": torch.tensor(float(d["step_before"])), "exp_avg": torch.from_numpy(d["m_before"].copy()),
                    "exp_avg_sq": torch.from_numpy(d["v_before"].copy())}
    gm = types.SimpleNamespace(_opacity=p, optimizer=opt)
    reset_opacity_nonvisible(gm, [torch.from_numpy(d["filter0"]), torch.from_numpy(d["filter1"])])
    assert isinstance(gm._opacity, torch.nn.Parameter) and gm._opacity is opt.param_groups[1]["params"][0] and gm._opacity is not p
    assert np.array_equal(gm._opacity.detach().numpy(), d["opacity_after"])
    st = opt.state[gm._opacity]
    assert np.array_equal(st["exp_avg"].numpy(), d["m_after"]) and np.array_equal(st["exp_avg_sq"].numpy(), d["v_after"])
    assert float(st["step"]) == float(d["step_after"]) == float(d["step_before"]) and not d["m_after"].any()
    seen = d["filter0"] | d["filter1"]
    assert np.allclose(d["opacity_after"][~seen], np.log(0.4 / 0.6), atol=1e-6)       # the reset value
    assert (d["opacity_after"][seen] > 0).all() and (d["opacity_after"][seen] < 1).all()  # the reference's quirk: sigmoid kept as logit
