"""Fused parameter activations + SH / feature packing on the GPU:
  * against the fixture recorded from the reference's own render() and autograd
    (tests/golden/activations.npz),
  * against the numpy oracle (oracle/activations.py) on larger seeded inputs,
  * splatloc_amd.fused.render against the same frame composed from torch ops + the rasterizer.
Tolerances: activations within a few ulp (1e-6 relative; expf / division are the accurate
ones); gradients 1e-5 of the tensor's scale.
"""
import types

import numpy as np
import pytest
import torch

from oracle import activations as act
from tests.test_oracle_activations import CASES, load_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _leaf(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).requires_grad_(True)


def _close(got, ref, what, rtol=1e-5):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    if ref.size == 0:
        return
    scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(got - ref).max()
    assert err <= rtol * scale + 1e-7, (what, err, scale)


@pytest.mark.parametrize("name", CASES)
def test_against_reference_fixture(name):
    from splatloc_amd.fused import activate_pack
    c = load_case(name)
    deg = int(c["active_sh_degree"])
    xyz, f_dc, f_rest = _leaf(c["raw_xyz"]), _leaf(c["raw_f_dc"]), _leaf(c["raw_f_rest"])
    scaling, rotation, opacity, kp = (_leaf(c["raw_" + k]) for k in ("scaling", "rotation", "opacity", "kp_score"))
    campos = torch.from_numpy(c["campos"]).to(DEV)
    scales, rotations, opacities, colors = activate_pack(xyz, f_dc, f_rest, scaling, rotation, opacity, extra=kp,
                                                         campos=campos, active_sh_degree=deg)
    np.testing.assert_allclose(scales.detach().cpu().numpy(), c["out_scales"], rtol=1e-6, atol=0)
    np.testing.assert_allclose(rotations.detach().cpu().numpy(), c["out_rotations"], rtol=0, atol=3e-7)
    np.testing.assert_allclose(opacities.detach().cpu().numpy(), c["out_opacities"], rtol=0, atol=3e-7)
    np.testing.assert_allclose(colors.detach().cpu().numpy(), c["out_colors_precomp"], rtol=0, atol=3e-6)
    G = {k: torch.from_numpy(c["G_" + k]).to(DEV) for k in ("means3D", "colors_precomp", "opacities", "scales", "rotations")}
    loss = (xyz * G["means3D"]).sum() + (colors * G["colors_precomp"]).sum() + (opacities * G["opacities"]).sum() \
        + (scales * G["scales"]).sum() + (rotations * G["rotations"]).sum()
    loss.backward()
    for leaf, key in ((xyz, "xyz"), (f_dc, "f_dc"), (scaling, "scaling"), (rotation, "rotation"),
                      (opacity, "opacity"), (kp, "kp_score")):
        _close(leaf.grad.cpu().numpy(), c["grad_" + key], key, rtol=3e-5)
    if c["raw_f_rest"].shape[1]:
        _close(f_rest.grad.cpu().numpy(), c["grad_f_rest"], "f_rest", rtol=3e-5)
    else:
        assert f_rest.grad is None or f_rest.grad.numel() == 0


@pytest.mark.parametrize("P,K,deg,SC,E", [(50_000, 1, 0, 3, 1), (20_000, 16, 3, 3, 0), (20_000, 16, 1, 1, 32),
                                          (1, 4, 1, 3, 2), (4097, 9, 2, 3, 1)])
def test_against_oracle(P, K, deg, SC, E):
    from splatloc_amd.fused import activate_pack
    g = torch.Generator().manual_seed(P + K)
    raw = dict(xyz=torch.randn(P, 3, generator=g) * 3, f_dc=torch.randn(P, 1, 3, generator=g),
               f_rest=torch.randn(P, K - 1, 3, generator=g), scaling=torch.randn(P, SC, generator=g) - 3,
               rotation=torch.randn(P, 4, generator=g), opacity=2 * torch.randn(P, 1, generator=g))
    extra = torch.rand(P, E, generator=g) if E else None
    campos = torch.tensor([0.3, -0.2, -4.0])
    leaves = {k: v.clone().to(DEV).requires_grad_(True) for k, v in raw.items()}
    ex = extra.clone().to(DEV).requires_grad_(True) if E else None
    outs = activate_pack(leaves["xyz"], leaves["f_dc"], leaves["f_rest"] if K > 1 else None, leaves["scaling"],
                         leaves["rotation"], leaves["opacity"], extra=ex, campos=campos.to(DEV), active_sh_degree=deg)
    npr = {k: v.numpy() for k, v in raw.items()}
    ref = act.forward(npr["xyz"], npr["f_dc"], npr["f_rest"], npr["scaling"], npr["rotation"], npr["opacity"],
                      extra.numpy() if E else None, campos.numpy(), deg)
    for t, k in zip(outs, ("scales", "rotations", "opacities", "colors")):
        np.testing.assert_allclose(t.detach().cpu().numpy(), ref[k], rtol=2e-6, atol=3e-6)
    Gs = [torch.randn(t.shape, generator=g) for t in outs]
    sum((t * G.to(DEV)).sum() for t, G in zip(outs, Gs)).backward()
    rb = act.backward(npr["xyz"], npr["f_dc"], npr["f_rest"], npr["scaling"], npr["rotation"], npr["opacity"],
                      extra.numpy() if E else None, campos.numpy(), deg, *[G.numpy() for G in Gs])
    _close(leaves["scaling"].grad.cpu().numpy(), rb["d_scaling"], "scaling", 3e-5)
    _close(leaves["rotation"].grad.cpu().numpy(), rb["d_rotation"], "rotation", 3e-5)
    _close(leaves["opacity"].grad.cpu().numpy(), rb["d_opacity"], "opacity", 3e-5)
    _close(leaves["f_dc"].grad.cpu().numpy(), rb["d_f_dc"], "f_dc", 3e-5)
    if K > 1:
        _close(leaves["f_rest"].grad.cpu().numpy(), rb["d_f_rest"], "f_rest", 3e-5)
    if deg > 0:
        _close(leaves["xyz"].grad.cpu().numpy(), rb["d_xyz"], "xyz", 1e-4)
    else:
        assert leaves["xyz"].grad is None
    if E:
        _close(ex.grad.cpu().numpy(), rb["d_extras"], "extras", 1e-6)


def _model(P, max_deg, active, seed, W, H):
    """A stand-in with the attributes render() reads from GaussianModel / Camera."""
    from splatloc_amd.camera import PinholeCamera
    g = torch.Generator().manual_seed(seed)
    K = (max_deg + 1) ** 2
    z = 0.8 + 4.0 * torch.rand(P, generator=g)
    xyz = torch.stack([(2 * torch.rand(P, generator=g) - 1) * z, (2 * torch.rand(P, generator=g) - 1) * 0.6 * z, z], 1)
    p = lambda t: t.to(DEV).requires_grad_(True)  # noqa: E731
    pc = types.SimpleNamespace(
        _xyz=p(xyz), _features_dc=p(0.5 * torch.randn(P, 1, 3, generator=g)),
        _features_rest=p(0.3 * torch.randn(P, K - 1, 3, generator=g)),
        _scaling=p(torch.log(0.04 * torch.exp(0.4 * torch.randn(P, 3, generator=g)))),
        _rotation=p(torch.randn(P, 4, generator=g)), _opacity=p(1.5 * torch.randn(P, 1, generator=g)),
        _kp_score=p(torch.rand(P, 1, generator=g)), active_sh_degree=active, max_sh_degree=max_deg)
    cam = PinholeCamera(W, H, W / 2.0, W / 2.0, (W - 1) / 2.0, (H - 1) / 2.0, torch.eye(3),
                        torch.tensor([0.05, -0.02, 0.1])).to(DEV)
    return pc, cam


def _composed_render(cam, pc, bg):
    """The reference's render() arithmetic with plain torch ops (gaussian_renderer/__init__.py:59-126)."""
    from splatloc_amd import GaussianRasterizationSettings, GaussianRasterizer
    from splatloc_amd.fused import math
    rs = GaussianRasterizationSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5),
                                       math.tan(cam.FoVy * 0.5), bg, 1.0, cam.world_view_transform,
                                       cam.full_proj_transform, pc.active_sh_degree, cam.camera_center, False, False)
    feats = torch.cat((pc._features_dc, pc._features_rest), dim=1)
    shs_view = feats.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
    d = pc._xyz - cam.camera_center.repeat(feats.shape[0], 1)
    d = d / d.norm(dim=1, keepdim=True)
    # eval_sh with torch ops: basis polynomials written out through autograd-friendly torch
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    res = act.C0 * shs_view[..., 0]
    if pc.active_sh_degree > 0:
        res = res - act.C1 * y * shs_view[..., 1] + act.C1 * z * shs_view[..., 2] - act.C1 * x * shs_view[..., 3]
    if pc.active_sh_degree > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = res + act.C2[0] * xy * shs_view[..., 4] + act.C2[1] * yz * shs_view[..., 5] \
            + act.C2[2] * (2.0 * zz - xx - yy) * shs_view[..., 6] + act.C2[3] * xz * shs_view[..., 7] \
            + act.C2[4] * (xx - yy) * shs_view[..., 8]
    rgb = torch.clamp_min(res + 0.5, 0.0)
    means2D = torch.zeros_like(pc._xyz, requires_grad=True)
    out = GaussianRasterizer(raster_settings=rs)(
        means3D=pc._xyz, means2D=means2D, shs=None, colors_precomp=torch.cat((rgb, pc._kp_score), dim=1),
        opacities=torch.sigmoid(pc._opacity), scales=torch.exp(pc._scaling),
        rotations=torch.nn.functional.normalize(pc._rotation), cov3D_precomp=None)
    return out, means2D


@pytest.mark.parametrize("max_deg,active", [(0, 0), (2, 2)])
def test_fused_render_matches_composed_render(max_deg, active):
    from splatloc_amd.fused import render
    W, H = 160, 112
    pc, cam = _model(3000, max_deg, active, seed=5 + max_deg, W=W, H=H)
    bg = torch.zeros(3, device=DEV)
    pipe = types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False)
    g = torch.Generator().manual_seed(1)
    w_img, w_d = torch.rand(4, H, W, generator=g).to(DEV), torch.rand(1, H, W, generator=g).to(DEV)
    names = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity", "_kp_score")

    (color, depth, alpha, radii), m2 = _composed_render(cam, pc, bg)
    ((color * w_img).sum() + (depth * w_d).sum()).backward()
    ref = {n: (getattr(pc, n).grad.clone() if getattr(pc, n).grad is not None else None) for n in names}
    ref_m2 = m2.grad.clone()
    for n in names:
        getattr(pc, n).grad = None

    out = render(cam, pc, pipe, bg)
    assert sorted(out) == ["depth", "kp_prob", "opacity", "radii", "render", "viewspace_points", "visibility_filter"]
    assert out["render"].shape == (3, H, W) and out["kp_prob"].shape == (H, W) and out["depth"].shape == (1, H, W)
    assert torch.equal(out["radii"], radii) and torch.equal(out["visibility_filter"], radii > 0)
    full = torch.cat((out["render"], out["kp_prob"][None]), dim=0)
    # ulp-level differences of the activations can flip an alpha >= 1/255 test at isolated pixels
    # (a jump of <= 1/255 * T * colour), so: nearly all pixels to 2e-5, every pixel to 1/255
    diff = (full - color).detach().abs()
    assert float((diff > 2e-5).float().mean()) < 1e-4 and float(diff.max()) <= 1.0 / 255.0
    ddiff = (out["depth"] - depth).detach().abs()
    assert float((ddiff > 2e-5 * float(depth.detach().abs().max())).float().mean()) < 1e-4
    adiff = (out["opacity"] - alpha).detach().abs()
    assert float((adiff > 2e-5).float().mean()) < 1e-4 and float(adiff.max()) <= 1.0 / 255.0
    ((full * w_img).sum() + (out["depth"] * w_d).sum()).backward()
    for n in names:
        got = getattr(pc, n).grad
        if ref[n] is None or ref[n].numel() == 0:
            assert got is None or got.numel() == 0 or float(got.abs().max()) == 0.0
            continue
        scale = float(ref[n].abs().max())
        assert float((got - ref[n]).abs().max()) <= 3e-3 * scale + 1e-9, n
    vs = out["viewspace_points"].grad
    assert vs is not None and float((vs - ref_m2).abs().max()) <= 3e-3 * float(ref_m2.abs().max()) + 1e-9


def test_render_variants_and_errors():
    from splatloc_amd.fused import activate_pack, render
    pc, cam = _model(1500, 1, 1, seed=9, W=96, H=64)
    bg = torch.zeros(3, device=DEV)
    base = render(cam, pc, types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False), bg)
    # in-kernel SH (convert_SHs_python False): RGB identical up to rounding, 3 channels only
    shs = render(cam, pc, types.SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False), bg)
    assert float((shs["render"] - base["render"]).abs().max()) <= 2e-5
    # python covariance path
    cov = render(cam, pc, types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=True), bg)
    assert float((cov["render"] - base["render"]).abs().max()) <= 2e-4
    assert torch.equal(cov["radii"], base["radii"])
    # override_color is ignored exactly as in the reference (gaussian_renderer/__init__.py:83-98: dead branch)
    oc = torch.rand(1500, 3, device=DEV)
    ov = render(cam, pc, types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False), bg, override_color=oc)
    assert torch.equal(ov["render"], base["render"]) and torch.equal(ov["kp_prob"], base["kp_prob"])
    # mask: the masked subset; with convert_SHs_python False 3 SH channels and kp_prob = channel 2 (reference :104-115)
    mask = torch.arange(1500, device=DEV) % 3 != 0
    out = render(cam, pc, types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False), bg, mask=mask)
    assert out["radii"].shape[0] == int(mask.sum())
    outs = render(cam, pc, types.SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False), bg, mask=mask)
    assert torch.equal(outs["kp_prob"], outs["render"][2]) and float((outs["render"] - out["render"]).abs().max()) <= 2e-5
    # empty model -> None, like the reference
    empty = types.SimpleNamespace(_xyz=torch.empty(0, 3, device=DEV))
    assert render(cam, empty, None, bg) is None
    with pytest.raises(RuntimeError):
        activate_pack(pc._xyz, pc._features_dc, None, pc._scaling, pc._rotation, pc._opacity, campos=None,
                      active_sh_degree=1)
    with pytest.raises(RuntimeError):
        activate_pack(pc._xyz.cpu(), pc._features_dc.cpu(), None, pc._scaling.cpu(), pc._rotation.cpu(),
                      pc._opacity.cpu())
