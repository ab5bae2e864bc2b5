"""Device-side densification statistics against the fixture recorded from the reference."""
import numpy as np
import pytest
import torch

from tests.test_oracle_stats import GOLD

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_densification_stats_match_reference():
    from splatloc_amd.densify import add_densification_stats
    d = np.load(GOLD)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
    a, n, m = t(d["accum0"]), t(d["denom0"]), t(d["max_radii0"])
    for v in range(3):
        add_densification_stats(t(d[f"grad{v}"]), t(d[f"radii{v}"]), a, n, m)
        np.testing.assert_allclose(a.cpu().numpy(), d[f"accum{v + 1}"], rtol=1e-6, atol=1e-9)
        assert np.array_equal(n.cpu().numpy(), d[f"denom{v + 1}"])
        assert np.array_equal(m.cpu().numpy(), d[f"max_radii{v + 1}"])
    with pytest.raises(RuntimeError):
        add_densification_stats(t(d["grad0"]), t(d["radii0"]).float(), a, n, m)
    with pytest.raises(RuntimeError):
        add_densification_stats(t(d["grad0"]).cpu(), t(d["radii0"]), a, n, m)


# ---------------------------------------------------------------------------------------------------
# densify / clone / split / prune + Adam against tests/golden/densify.npz — recorded
# from the reference's own GaussianModel.densify_and_prune and torch.optim.Adam
# ---------------------------------------------------------------------------------------------------
import os
import types

from oracle import densify as od

G = od.GROUPS
DENS = os.path.join(os.path.dirname(GOLD), "densify.npz")
ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
        "marker": "_marker", "kp_score": "_kp_score", "scaling": "_scaling", "rotation": "_rotation"}


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _model_from_fixture(d, pre, adam_cls):
    """A stand-in with the attribute / optimizer layout of the reference's GaussianModel after training_setup
    (gaussian_model.py:250-300): 8 named groups, lr per group, eps 1e-15."""
    gm = types.SimpleNamespace()
    groups = []
    for k in G:
        p = torch.nn.Parameter(_t(d[pre + k]).requires_grad_(True))
        setattr(gm, ATTR[k], p)
        groups.append({"params": [p], "lr": float(d[f"{pre}lr_{k}"]), "name": k})
    gm.optimizer = adam_cls(groups, lr=0.0, eps=1e-15)
    return gm


def _run_steps(gm, d, pre, n, first_iteration, lrs_after):
    for k in range(n):
        for name in G:
            key = f"{pre}grad{k}_{name}"
            getattr(gm, ATTR[name]).grad = _t(d[key]) if key in d.files else None
        gm.optimizer.step()
        gm.optimizer.zero_grad(set_to_none=True)
        for grp in gm.optimizer.param_groups:     # update_learning_rate (gaussian_model.py:311-325)
            if grp["name"] == "xyz":
                grp["lr"] = float(d[f"{pre}xyz_lr_after{k}"])


def _check_model(gm, d, pre, rtol, atol, step):
    for k in G:
        got = getattr(gm, ATTR[k]).detach().cpu().numpy()
        assert got.shape == d[pre + k].shape, k
        np.testing.assert_allclose(got, d[pre + k], rtol=rtol, atol=atol, err_msg=k)
    for grp in gm.optimizer.param_groups:
        st = gm.optimizer.state.get(grp["params"][0], {})
        k = grp["name"]
        if f"{pre}m_{k}" in d.files:
            np.testing.assert_allclose(st["exp_avg"].cpu().numpy(), d[f"{pre}m_{k}"], rtol=1e-5, atol=1e-10, err_msg=k)
            np.testing.assert_allclose(st["exp_avg_sq"].cpu().numpy(), d[f"{pre}v_{k}"], rtol=1e-5, atol=1e-14, err_msg=k)
            assert float(st["step"]) == step == float(d[f"{pre}step_{k}"])
        else:
            assert len(st) == 0, f"{k}: a group without gradient must not get optimizer state"


@pytest.mark.parametrize("case", ["plain_", "reg_"])
def test_fused_adam_and_densify_match_reference(case):
    """The whole sequence of the fixture on the device: 3 fused-Adam steps, densify_and_prune on the
    reference-shaped model object (injected split noise), 2 more steps on the re-sized model."""
    from splatloc_amd.densify import densify_and_prune
    from splatloc_amd.optim import Adam
    d = np.load(DENS)
    extent, pd, max_grad, min_opacity, size_thr = (float(v) for v in d[case + "hyper"])
    gm = _model_from_fixture(d, case + "s0_", Adam)
    _run_steps(gm, d, case + "a_", 3, 1, None)
    _check_model(gm, d, case + "s1_", rtol=2e-6, atol=1e-7, step=3.0)
    assert len(gm.optimizer.state.get(gm._marker, {})) == 0 and gm._features_rest.shape[1:] == (0, 3)
    # densify from the REFERENCE's pre-state so that rounding of the 3 steps cannot move a threshold decision
    gm = _model_from_fixture(d, case + "s1_", Adam)
    for grp in gm.optimizer.param_groups:
        k = grp["name"]
        if f"{case}s1_m_{k}" in d.files:
            gm.optimizer.state[grp["params"][0]] = {"step": torch.tensor(float(d[f"{case}s1_step_{k}"])),
                                                    "exp_avg": _t(d[f"{case}s1_m_{k}"]),
                                                    "exp_avg_sq": _t(d[f"{case}s1_v_{k}"])}
    gm.xyz_gradient_accum, gm.denom = _t(d[case + "accum_in"]), _t(d[case + "denom_in"])
    gm.max_radii2D = _t(d[case + "max_radii_in"])
    gm.percent_dense, gm.primitive_reg = pd, case == "reg_"
    n = densify_and_prune(gm, max_grad, min_opacity, extent, size_thr, unit_noise=_t(d[case + "unit_noise"]))
    assert n == d[case + "s2_xyz"].shape[0] == gm._xyz.shape[0]
    for k in G:   # copies are exact; the split's xyz / scaling pass through sincos-free float math (rotation, log/exp)
        tol = dict(rtol=3e-6, atol=3e-6) if k in ("xyz", "scaling") else dict(rtol=0, atol=0)
        np.testing.assert_allclose(getattr(gm, ATTR[k]).detach().cpu().numpy(), d[case + "s2_" + k], err_msg=k, **tol)
        assert isinstance(getattr(gm, ATTR[k]), torch.nn.Parameter) and getattr(gm, ATTR[k]).requires_grad
    for grp in gm.optimizer.param_groups:
        k, p = grp["name"], grp["params"][0]
        assert p is getattr(gm, ATTR[k])
        if f"{case}s2_m_{k}" in d.files:
            st = gm.optimizer.state[p]
            assert np.array_equal(st["exp_avg"].cpu().numpy(), d[f"{case}s2_m_{k}"]), k
            assert np.array_equal(st["exp_avg_sq"].cpu().numpy(), d[f"{case}s2_v_{k}"]), k
            assert float(st["step"]) == 3.0
    assert gm.xyz_gradient_accum.shape == (n, 1) and not gm.xyz_gradient_accum.any() and not gm.max_radii2D.any()
    # the reference's lr after its 3 steps, then 2 more steps
    for grp in gm.optimizer.param_groups:
        grp["lr"] = float(d[f"{case}s2_lr_{grp['name']}"])
    gm2 = gm
    for k in G:     # continue from the reference's exact s2 parameters (xyz / scaling differ in the last ulp)
        with torch.no_grad():
            getattr(gm2, ATTR[k]).copy_(_t(d[case + "s2_" + k]))
    _run_steps(gm2, d, case + "b_", 2, 4, None)
    _check_model(gm2, d, case + "s3_", rtol=2e-6, atol=1e-7, step=5.0)


def test_densify_tensor_entry_sources_and_oracle():
    """densify_tensors vs the numpy oracle on a larger seeded model (isotropic scaling, kp width 2): same row
    provenance (source row, kind) and values."""
    from splatloc_amd.densify import densify_tensors
    rng = np.random.default_rng(3)
    P = 50_000
    par = dict(xyz=rng.normal(size=(P, 3)), f_dc=rng.random((P, 1, 3)), f_rest=np.zeros((P, 0, 3)),
               opacity=rng.normal(size=(P, 1)) * 2, marker=(rng.random((P, 1)) < 0.3) * rng.random((P, 1)),
               kp_score=rng.random((P, 2)), scaling=np.log(0.06) + 0.8 * rng.normal(size=(P, 3)),
               rotation=rng.normal(size=(P, 4)))
    par = {k: v.astype(np.float32) for k, v in par.items()}
    m = {k: rng.normal(size=par[k].shape).astype(np.float32) for k in G if k != "marker"}
    v = {k: rng.random(par[k].shape).astype(np.float32) for k in G if k != "marker"}
    accum = (rng.random((P, 1)) * 0.002).astype(np.float32)
    denom = rng.integers(0, 4, (P, 1)).astype(np.float32)
    unit = rng.normal(size=(2, P, 3)).astype(np.float32)
    ref_p, ref_s, src, kind = od.densify_and_prune(par, {k: dict(m=m[k], v=v[k], step=7.0) for k in m}, accum, denom, unit,
                                                   0.0002, 0.3, 6.0, 20.0, 0.01, primitive_reg=True)
    out_p, out_m, out_v, srow, skind = densify_tensors(
        {k: _t(a) for k, a in par.items()}, {k: _t(a) for k, a in m.items()}, {k: _t(a) for k, a in v.items()},
        _t(accum), _t(denom), 0.0002, 0.3, 6.0, 20.0, 0.01, True, unit_noise=_t(unit), return_sources=True)
    assert out_p["xyz"].shape[0] == ref_p["xyz"].shape[0]
    assert np.array_equal(srow.cpu().numpy(), src) and np.array_equal(skind.cpu().numpy(), kind)
    for k in G:
        tol = dict(rtol=3e-6, atol=3e-6) if k in ("xyz", "scaling") else dict(rtol=0, atol=0)
        np.testing.assert_allclose(out_p[k].cpu().numpy(), ref_p[k], err_msg=k, **tol)
    for k in m:
        assert np.array_equal(out_m[k].cpu().numpy(), ref_s[k]["m"]) and np.array_equal(out_v[k].cpu().numpy(), ref_s[k]["v"])
    assert "marker" not in out_m


def test_split_draws_are_seeded_and_normal():
    """Without an injected table the split children come from the counter-based generator: identical for the same
    (seed, draw_id) — what lets data-parallel replicas densify identically without a broadcast — different
    otherwise, and standard normal: offsets in the Gaussian's frame have mean 0 and the stored scale as std."""
    from splatloc_amd.densify import densify_tensors
    P = 200_000
    g = torch.Generator().manual_seed(1)
    par = dict(xyz=torch.zeros(P, 3), f_dc=torch.rand(P, 1, 3, generator=g), f_rest=torch.zeros(P, 0, 3),
               opacity=torch.full((P, 1), 3.0), marker=torch.zeros(P, 1), kp_score=torch.rand(P, 1, generator=g),
               scaling=torch.log(torch.tensor([0.1, 0.2, 0.4])).repeat(P, 1), rotation=torch.tensor([[2.0, 0, 0, 0]]).repeat(P, 1))
    par = {k: t.contiguous().to(DEV) for k, t in par.items()}
    accum, denom = torch.full((P, 1), 1.0, device=DEV), torch.ones(P, 1, device=DEV)
    run = lambda seed, did: densify_tensors(par, {}, {}, accum, denom, 0.0002, 0.005, 100.0, 0, 0.001, False,  # noqa: E731
                                            seed=seed, draw_id=did, return_sources=True)
    a, _, _, srow, kind = run(11, 0)
    b = run(11, 0)[0]
    c = run(11, 1)[0]
    e = run(12, 0)[0]
    assert a["xyz"].shape[0] == 2 * P and bool((kind >= 2).all())       # every row split, parents removed
    assert torch.equal(a["xyz"], b["xyz"]) and not torch.equal(a["xyz"], c["xyz"]) and not torch.equal(a["xyz"], e["xyz"])
    off = a["xyz"].double()                                            # identity rotation, parents at the origin
    assert float(off.mean(0).abs().max()) < 3e-3
    np.testing.assert_allclose(off.std(0).cpu().numpy(), [0.1, 0.2, 0.4], rtol=1e-2)
    k4 = ((off / torch.tensor([0.1, 0.2, 0.4], device=DEV, dtype=torch.float64)) ** 4).mean(0)   # kurtosis 3
    np.testing.assert_allclose(k4.cpu().numpy(), [3.0, 3.0, 3.0], rtol=5e-2)
    np.testing.assert_allclose(a["scaling"][0].exp().cpu().numpy(), np.array([0.1, 0.2, 0.4]) / 1.6, rtol=1e-5)
    # the two copies of one parent differ
    assert not torch.equal(a["xyz"][:P], a["xyz"][P:])


def test_adam_key_gate_and_missing_grads():
    """The key-primitive freeze (train_gaussians.py:231-234) inside the fused step == torch.optim.Adam with the
    gated rows' gradient zeroed by hand; parameters without .grad are skipped."""
    from splatloc_amd.optim import Adam
    g = torch.Generator().manual_seed(9)
    P = 10_001
    xyz0, sc0 = torch.randn(P, 3, generator=g), torch.randn(P, 3, generator=g)
    marker = ((torch.rand(P, 1, generator=g) < 0.4).float() * torch.rand(P, 1, generator=g)).to(DEV)
    mk = lambda: [torch.nn.Parameter(xyz0.clone().to(DEV)), torch.nn.Parameter(sc0.clone().to(DEV)),  # noqa: E731
                  torch.nn.Parameter(marker.clone())]
    pa, pb = mk(), mk()
    groups = lambda ps: [{"params": [ps[0]], "lr": 1e-2, "name": "xyz"}, {"params": [ps[1]], "lr": 3e-3, "name": "scaling"},  # noqa: E731
                         {"params": [ps[2]], "lr": 5e-2, "name": "marker"}]
    fused = Adam(groups(pa), lr=0.0, eps=1e-15)
    ref = torch.optim.Adam(groups(pb), lr=0.0, eps=1e-15)
    fused.set_key_gate(marker, 0.005)
    for it in range(4):
        gx, gs = (torch.randn(P, 3, generator=g) * 1e-3).to(DEV), (torch.randn(P, 3, generator=g) * 1e-3).to(DEV)
        pa[0].grad, pa[1].grad, pa[2].grad = gx.clone(), gs.clone(), None
        gz = gx.clone()
        gz[marker.squeeze() > 0.005] = 0
        pb[0].grad, pb[1].grad, pb[2].grad = gz, gs.clone(), None
        fused.step()
        ref.step()
    for a, b in zip(pa[:2], pb[:2]):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)
    assert torch.equal(pa[2], pb[2]) and len(fused.state.get(pa[2], {})) == 0


def test_adam_unaligned_tensors_take_the_scalar_kernel_and_agree():
    """The fused step reads and writes 16 bytes at a time when every tensor of every group is 16-byte aligned (torch allocations
    are), and element by element otherwise (densify.hip): the same per-element arithmetic — a model whose tensors are views one
    float into their storage ends bit-identical to an aligned twin, gate and partial last quad included."""
    from splatloc_amd.optim import Adam
    g = torch.Generator().manual_seed(19)
    P = 5_003
    xyz0, op0 = torch.randn(P, 3, generator=g), torch.randn(P, 1, generator=g)
    marker = ((torch.rand(P, 1, generator=g) < 0.4).float() * torch.rand(P, 1, generator=g)).to(DEV)

    def shifted(t):     # the same values, one float into a larger buffer: 4-byte aligned only
        buf = torch.empty(t.numel() + 1, device=DEV)
        v = buf[1:].view(t.shape)
        v.copy_(t)
        assert v.data_ptr() % 16 != 0 and v.is_contiguous()
        return v

    pa = [torch.nn.Parameter(xyz0.clone().to(DEV)), torch.nn.Parameter(op0.clone().to(DEV))]
    pb = [torch.nn.Parameter(shifted(xyz0.to(DEV))), torch.nn.Parameter(shifted(op0.to(DEV)))]
    mk = lambda ps: Adam([{"params": [ps[0]], "lr": 1e-2, "name": "xyz"}, {"params": [ps[1]], "lr": 5e-2, "name": "opacity"}], lr=0.0, eps=1e-15)  # noqa: E731
    oa, ob = mk(pa), mk(pb)
    oa.set_key_gate(marker, 0.005)
    ob.set_key_gate(marker, 0.005)
    for it in range(3):
        gx, go = (torch.randn(P, 3, generator=g) * 1e-3).to(DEV), (torch.randn(P, 1, generator=g) * 1e-3).to(DEV)
        pa[0].grad, pa[1].grad = gx.clone(), go.clone()
        pb[0].grad, pb[1].grad = shifted(gx), shifted(go)
        oa.step()
        ob.step()
    for a, b in zip(pa, pb):
        assert torch.equal(a.detach(), b.detach())
        assert torch.equal(oa.state[a]["exp_avg_sq"], ob.state[b]["exp_avg_sq"])


def test_adam_step_carries_the_max_radii_line():
    """`set_radii_update`: the next fused step also performs SplatLoc.color_refinement's statistics line (train_gaussians.py:
    293-294) — same parameters as a plain step, the same max_radii2D as the statistics launch; also with no gradient at all."""
    from splatloc_amd.densify import add_densification_stats_window
    from splatloc_amd.optim import Adam
    g = torch.Generator().manual_seed(23)
    P = 7_777
    x0 = torch.randn(P, 3, generator=g)
    radii = (torch.randint(0, 40, (P,), generator=g) * (torch.rand(P, generator=g) < 0.6)).to(torch.int32).to(DEV)
    m0 = (torch.rand(P, generator=g) * 30).to(DEV)
    pa, pb = torch.nn.Parameter(x0.clone().to(DEV)), torch.nn.Parameter(x0.clone().to(DEV))
    oa = Adam([{"params": [pa], "lr": 1e-2, "name": "xyz"}], lr=0.0, eps=1e-15)
    ob = Adam([{"params": [pb], "lr": 1e-2, "name": "xyz"}], lr=0.0, eps=1e-15)
    ma, mb = m0.clone(), m0.clone()
    for it in range(3):
        gx = (torch.randn(P, 3, generator=g) * 1e-3).to(DEV)
        pa.grad, pb.grad = gx.clone(), gx.clone()
        oa.set_radii_update(radii, ma)
        oa.step()
        add_densification_stats_window(None, [radii], None, None, mb)
        ob.step()
    assert torch.equal(pa.detach(), pb.detach()) and torch.equal(ma, mb)
    assert torch.equal(ma, torch.where(radii > 0, torch.maximum(m0, radii.float()), m0))
    pa.grad = None
    r2 = (radii + 50) * (radii > 0)
    oa.set_radii_update(r2.to(torch.int32), ma)
    oa.step()                                   # no gradient anywhere: the statistics alone
    assert torch.equal(ma, torch.where(radii > 0, r2.float(), m0)) and torch.equal(pa.detach(), pb.detach())


def test_isotropic_loss_matches_reference_expression():
    from splatloc_amd.losses import isotropic_loss
    g = torch.Generator().manual_seed(4)
    for SC in (3, 1):
        P = 20_003
        raw = (torch.randn(P, SC, generator=g) * 0.5 + np.log(0.02)).to(DEV).requires_grad_(True)
        marker = ((torch.rand(P, 1, generator=g) < 0.3).float() * torch.rand(P, 1, generator=g) * 0.9).to(DEV)
        loss = isotropic_loss(torch.exp(raw), marker)
        loss.backward()
        raw2 = raw.detach().clone().requires_grad_(True)
        scaling = torch.exp(raw2)
        mask = marker.squeeze() > 0.005                                 # train_gaussians.py:223-226
        ref = torch.abs(scaling.mean(dim=1).view(-1, 1)[mask] / (0.02 * (1 - marker[mask])) - 1).mean()
        ref.backward()
        np.testing.assert_allclose(float(loss), float(ref), rtol=1e-5)
        np.testing.assert_allclose(raw.grad.cpu().numpy(), raw2.grad.cpu().numpy(), rtol=1e-4, atol=1e-9)
    z = isotropic_loss(torch.ones(10, 3, device=DEV), torch.zeros(10, 1, device=DEV))   # empty mask
    assert float(z) == 0.0
