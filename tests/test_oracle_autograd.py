"""Cross-checks oracle/splat_oracle.c (forward AND hand-written backward) against an
independent dense fp64 PyTorch restatement + torch.autograd (tests/torch_dense_ref.py)."""
import numpy as np
import pytest
import torch

from oracle import oracle
from splatloc_amd.synthetic import make_scene
from tests.torch_dense_ref import render_dense


def _run(P, W, H, C, seed, use_sh=False, deg=0, use_cov=False, mod=1.0, bgval=0.3):
    sc = make_scene(P, W, H, C, seed, scale_median=0.05)
    sc.opacities = sc.opacities.clamp(max=0.95)  # keep away from the 0.99 clamp's kink
    cam = sc.camera
    # a non-trivial pose so view/proj are dense
    ang = 0.2
    R = torch.tensor([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], dtype=torch.float32)
    from splatloc_amd.camera import PinholeCamera
    cam = PinholeCamera(W, H, cam.fx, cam.fy, cam.cx + 0.7, cam.cy - 0.3, R, torch.tensor([0.1, -0.05, 0.4]))
    g = torch.Generator().manual_seed(seed + 100)
    bg = torch.full((min(C, 3),), bgval)
    shs = 0.5 * torch.randn(P, 16, 3, generator=g) if use_sh else None
    cov = None
    if use_cov:
        L = torch.randn(P, 3, 3, generator=g) * 0.04
        S = L @ L.transpose(1, 2)
        cov = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1)
    st = oracle.Settings(H, W, cam.tanfovx, cam.tanfovy, scale_modifier=mod, sh_degree=deg)
    kw = {}
    if use_sh:
        kw["shs"] = shs.numpy()
    else:
        kw["colors_precomp"] = sc.features.numpy()
    if use_cov:
        kw["cov3D_precomp"] = cov.numpy()
    else:
        kw["scales"], kw["rotations"] = sc.scales.numpy(), sc.rotations.numpy()
    f = oracle.forward(st, bg.numpy(), sc.means3D.numpy(), sc.opacities.numpy(), cam.world_view_transform.numpy(),
                       cam.full_proj_transform.numpy(), cam.camera_center.numpy(), **kw)
    Cn = 3 if use_sh else C
    dcol = sc.dL_dcolor[:Cn] * (H * W)
    ddep = sc.dL_ddepth * (H * W)
    dalp = sc.dL_dalpha * (H * W)
    b = oracle.backward(f, dcol.numpy(), ddep.numpy(), dalp.numpy())

    d = torch.float64
    leaf = lambda t: None if t is None else t.to(d).clone().requires_grad_(True)  # noqa: E731
    m3, op = leaf(sc.means3D), leaf(sc.opacities)
    col, sh_t = (None, leaf(shs)) if use_sh else (leaf(sc.features), None)
    sca, rot, cv = (None, None, leaf(cov)) if use_cov else (leaf(sc.scales), leaf(sc.rotations), None)
    probe = torch.zeros(P, 2, dtype=d, requires_grad=True)
    Vm, PMm, cpos = leaf(cam.world_view_transform), leaf(cam.full_proj_transform), leaf(cam.camera_center)
    color, depth, alpha, radii = render_dense(
        H, W, cam.tanfovx, cam.tanfovy, bg, m3, op, Vm, PMm, cpos, colors_precomp=col, shs=sh_t, sh_degree=deg,
        scales=sca, rotations=rot, cov3D_precomp=cv, scale_modifier=mod, means2D_probe=probe)
    loss = (color * dcol.to(d)).sum() + (depth * ddep.to(d)).sum() + (alpha * dalp.to(d)).sum()
    loss.backward()
    return sc, f, b, dict(color=color, depth=depth, alpha=alpha, radii=radii, m3=m3, op=op, col=col, sh=sh_t,
                          sca=sca, rot=rot, cov=cv, probe=probe, V=Vm, PM=PMm, campos=cpos)


def _close(name, got, ref, rtol=2e-4, atol_scale=2e-5):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    tol = rtol * np.abs(ref) + atol_scale * np.abs(ref).max()
    bad = np.abs(got - ref) > tol
    assert not bad.any(), f"{name}: {bad.sum()}/{bad.size} off, worst {np.abs(got - ref).max():.3e} scale {np.abs(ref).max():.3e}"


@pytest.mark.parametrize("cfg", [
    dict(P=150, W=64, H=48, C=3, seed=3),
    dict(P=200, W=70, H=50, C=5, seed=4, mod=1.3),           # ragged image edge, C = 5, scale modifier
    dict(P=120, W=48, H=48, C=3, seed=5, use_sh=True, deg=3),
    dict(P=120, W=48, H=48, C=3, seed=6, use_sh=True, deg=1),
    dict(P=100, W=48, H=32, C=4, seed=7, use_cov=True),
])
def test_oracle_forward_and_backward_match_autograd(cfg):
    sc, f, b, t = _run(**cfg)
    assert (f["radii"] == t["radii"].numpy()).all()
    assert (f["radii"] > 0).sum() > 20
    np.testing.assert_allclose(f["color"], t["color"].detach().numpy(), atol=2e-5)
    np.testing.assert_allclose(f["depth"], t["depth"].detach().numpy(), atol=1e-4)
    np.testing.assert_allclose(f["alpha"], t["alpha"].detach().numpy(), atol=2e-5)
    _close("dL_dmeans3D", b["dL_dmeans3D"], t["m3"].grad.numpy())
    _close("dL_dmeans2D", b["dL_dmeans2D"][:, :2], t["probe"].grad.numpy())
    assert (b["dL_dmeans2D"][:, 2] == 0).all()
    _close("dL_dopacities", b["dL_dopacities"], t["op"].grad.numpy())
    if t["col"] is not None:
        _close("dL_dcolors", b["dL_dcolors"], t["col"].grad.numpy())
    else:
        _close("dL_dshs", b["dL_dshs"], t["sh"].grad.numpy())
    if t["cov"] is not None:
        _close("dL_dcov3D", b["dL_dcov3D"], t["cov"].grad.numpy())
    else:
        _close("dL_dscales", b["dL_dscales"], t["sca"].grad.numpy())
        _close("dL_drotations", b["dL_drotations"], t["rot"].grad.numpy())
    # pose-gradient extension: exact derivative w.r.t. the camera tensors
    _close("dL_dviewmatrix", b["dL_dviewmatrix"], t["V"].grad.numpy(), rtol=5e-4, atol_scale=5e-5)
    _close("dL_dprojmatrix", b["dL_dprojmatrix"], t["PM"].grad.numpy(), rtol=5e-4, atol_scale=5e-5)
    if t["sh"] is not None:
        _close("dL_dcampos", b["dL_dcampos"], t["campos"].grad.numpy(), rtol=5e-4, atol_scale=5e-5)
    else:
        assert (b["dL_dcampos"] == 0).all()


def test_oracle_omp_equals_single_thread():
    sc = make_scene(400, 96, 64, 4, 11, scale_median=0.04)
    from tests.helpers import oracle_forward, oracle_backward
    f1, f2 = oracle_forward(sc, omp=False), oracle_forward(sc, omp=True)
    for k in ("color", "depth", "alpha", "radii", "point_list", "ranges", "n_contrib"):
        assert np.array_equal(f1[k], f2[k]), k
    b1, b2 = oracle_backward(f1, sc, omp=False), oracle_backward(f2, sc, omp=True)
    for k in ("dL_dmeans3D", "dL_dcolors", "dL_dopacities", "dL_dscales", "dL_drotations"):
        np.testing.assert_allclose(b1[k], b2[k], rtol=1e-5, atol=1e-12)


def test_oracle_bin_order_is_tile_depth_index():
    sc = make_scene(500, 128, 96, 3, 12, scale_median=0.05)
    from tests.helpers import oracle_forward
    f = oracle_forward(sc)
    keys, vals = f["keys"], f["point_list"]
    assert (np.diff(keys.astype(np.uint64)) >= 0).all() if keys.size else True
    same = keys[1:] == keys[:-1]
    assert (vals[1:][same] > vals[:-1][same]).all()       # stable: ties by Gaussian index
    tiles = (keys >> np.uint64(32)).astype(np.int64)
    for t in np.unique(tiles)[:50]:
        s, e = f["ranges"][t]
        assert (tiles[s:e] == t).all() and (s == 0 or tiles[s - 1] != t) and (e == len(tiles) or tiles[e] != t)
    assert int(f["tiles_touched"].sum()) == f["num_rendered"] == len(vals)
