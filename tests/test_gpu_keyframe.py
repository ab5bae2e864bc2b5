"""Key-frame insertion on the device (splatloc_amd.keyframe + densify.extend_from_pcd) against tests/golden/keyframe.npz,
recorded from the reference's own GaussianModel.create_pcd_from_image / create_pcd_from_image_and_depth_score /
extend_from_pcd (gaussian_model.py:118-248) — the only caller of simple_knn.distCUDA2 — needs an MI355X."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "marker", "kp_score", "scaling", "rotation")
ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
        "marker": "_marker", "kp_score": "_kp_score", "scaling": "_scaling", "rotation": "_rotation"}
LR = {"xyz": 1.6e-4 * 6.0, "f_dc": 2.5e-3, "f_rest": 2.5e-3 / 20, "opacity": 5e-2, "marker": 5e-2, "kp_score": 5e-2,
      "scaling": 1e-3 * 6.0, "rotation": 1e-3}


def _model(d, pre, adam_cls, dev):
    gm = types.SimpleNamespace(max_sh_degree=0, active_sh_degree=0, isotropic=False, percent_dense=0.01, primitive_reg=True,
                               config={"Dataset": {"pcd_downsample": int(d["cfg"][0]), "point_size": float(d["cfg"][1]),
                                                   "adaptive_pointsize": bool(d["cfg"][2])}})
    par = lambda a: torch.nn.Parameter(torch.from_numpy(a).to(dev).contiguous().requires_grad_(True))  # noqa: E731
    for k in GROUPS:
        setattr(gm, ATTR[k], par(d[pre + k]))
    gm.optimizer = adam_cls([{"params": [getattr(gm, ATTR[k])], "lr": LR[k], "name": k} for k in GROUPS], lr=0.0, eps=1e-15)
    for grp in gm.optimizer.param_groups:
        k = grp["name"]
        if bool(d[f"{pre}has_state_{k}"]):
            gm.optimizer.state[grp["params"][0]] = {
                "step": torch.tensor(float(d[f"{pre}step_{k}"])), "exp_avg": torch.from_numpy(d[f"{pre}m_{k}"]).to(dev),
                "exp_avg_sq": torch.from_numpy(d[f"{pre}v_{k}"]).to(dev)}
    gm.xyz_gradient_accum = torch.from_numpy(d[pre + "accum"]).to(dev)
    gm.denom = torch.from_numpy(d[pre + "denom"]).to(dev)
    gm.max_radii2D = torch.from_numpy(d[pre + "max_radii"]).to(dev)
    return gm


def _cam(d, dev):
    fx, fy, cx, cy, W, H = (float(v) for v in d["intr"])
    return types.SimpleNamespace(fx=fx, fy=fy, cx=cx, cy=cy, W2C=torch.from_numpy(d["view_T"]).to(dev),
                                 original_image=torch.from_numpy(d["view_color"]).to(dev),
                                 kp_score=torch.from_numpy(d["view_kp"]).to(dev),
                                 exposure_a=torch.tensor([float(d["view_exposure"][0])], device=dev),
                                 exposure_b=torch.tensor([float(d["view_exposure"][1])], device=dev))


def test_distcuda2_through_its_reference_caller(golden_dir):
    """The point cloud the reference handed to distCUDA2 (gaussian_model.py:206) through the HIP kernel: bit-exact
    against the exact 3-NN the fixture was recorded with."""
    from simple_knn._C import distCUDA2
    d = np.load(os.path.join(golden_dir, "keyframe.npz"))
    out = distCUDA2(torch.from_numpy(d["knn_points"]).to(DEV))
    assert np.array_equal(out.cpu().numpy().view(np.uint32), d["knn_dist2"].view(np.uint32))


@pytest.mark.parametrize("adam", ["torch", "fused"])
def test_keyframe_insertion_matches_reference_recording(golden_dir, adam):
    from splatloc_amd import keyframe
    from splatloc_amd.optim import Adam as FusedAdam
    d = np.load(os.path.join(golden_dir, "keyframe.npz"))
    dev = torch.device(DEV)
    gm = _model(d, "before_", torch.optim.Adam if adam == "torch" else FusedAdam, dev)
    cam = _cam(d, dev)
    with torch.no_grad():
        tensors = keyframe.create_pcd_from_image(gm, cam, torch.from_numpy(d["view_depth"]).to(dev), sample_idx=d["sample_idx"])
    names = ("fused_point_cloud", "features", "scales", "rots", "opacities", "markers", "kp_scores")
    # 1 ulp: the float32 pose inverse and point size; RGB2SH's `/ C0`, which torch evaluates as `* (1 / C0)` on the GPU
    # (the reference's own device) and as a division on the CPU the fixture was recorded on
    tol = {"fused_point_cloud": dict(rtol=0, atol=2e-6), "scales": dict(rtol=0, atol=5e-6), "features": dict(rtol=3e-7, atol=1e-7)}
    for n, t in zip(names, tensors):
        ref = d["pcd_" + n]
        assert tuple(t.shape) == ref.shape and t.dtype == torch.float32, n
        np.testing.assert_allclose(t.cpu().numpy(), ref, err_msg=n, **tol.get(n, dict(rtol=0, atol=0)))
    assert int(tensors[0].shape[0]) == int(d["num_kp"]) + d["sample_idx"].shape[0]
    # scales = log(sqrt(clamp_min(distCUDA2(xyz), 1e-7) * point_size)) repeated on 3 axes: isotropic seeds
    assert torch.equal(tensors[2][:, 0], tensors[2][:, 1]) and torch.equal(tensors[2][:, 0], tensors[2][:, 2])

    # the append: feed the REFERENCE's recorded rows so that the comparison is exact
    from splatloc_amd.densify import extend_from_pcd
    rows = [torch.from_numpy(d["pcd_" + n]).to(dev) for n in names]
    old = {k: getattr(gm, ATTR[k]) for k in GROUPS}
    n = extend_from_pcd(gm, *rows)
    assert n == d["after_xyz"].shape[0] == d["before_xyz"].shape[0] + d["pcd_fused_point_cloud"].shape[0]
    for grp in gm.optimizer.param_groups:
        k = grp["name"]
        p = grp["params"][0]
        assert p is getattr(gm, ATTR[k]) and isinstance(p, torch.nn.Parameter) and p.requires_grad and p is not old[k]
        assert old[k] not in gm.optimizer.state
        assert np.array_equal(p.detach().cpu().numpy(), d["after_" + k]), k
        st = gm.optimizer.state.get(p, None)
        assert bool(d[f"after_has_state_{k}"]) == bool(st is not None and len(st)), k
        if st is not None and len(st):
            assert float(st["step"]) == float(d[f"after_step_{k}"])
            assert np.array_equal(st["exp_avg"].cpu().numpy(), d[f"after_m_{k}"]), k       # old rows kept, new rows zero
            assert np.array_equal(st["exp_avg_sq"].cpu().numpy(), d[f"after_v_{k}"]), k
    assert np.array_equal(gm.xyz_gradient_accum.cpu().numpy(), d["after_accum"])
    assert np.array_equal(gm.denom.cpu().numpy(), d["after_denom"])
    assert np.array_equal(gm.max_radii2D.cpu().numpy(), d["after_max_radii"])
    # the model keeps training: one optimizer step on the re-sized tensors
    for k in GROUPS:
        p = getattr(gm, ATTR[k])
        p.grad = None if k == "marker" else torch.full_like(p, 1e-3)
    gm.optimizer.step()
    assert torch.isfinite(gm._xyz).all()


def test_first_keyframe_into_an_empty_model():
    """The very first key-frame: an empty model whose optimizer has no state yet (SplatLoc.__init__ -> training_setup,
    then add_next_kf, train_gaussians.py:69,332)."""
    from splatloc_amd.densify import extend_from_pcd
    dev = torch.device(DEV)
    gm = types.SimpleNamespace(max_sh_degree=0)
    shapes = {"xyz": (0, 3), "f_dc": (0, 1, 3), "f_rest": (0, 0, 3), "opacity": (0, 1), "marker": (0, 1), "kp_score": (0, 1),
              "scaling": (0, 3), "rotation": (0, 4)}
    for k in GROUPS:
        setattr(gm, ATTR[k], torch.nn.Parameter(torch.empty(shapes[k], device=dev)))
    gm.optimizer = torch.optim.Adam([{"params": [getattr(gm, ATTR[k])], "lr": LR[k], "name": k} for k in GROUPS], lr=0.0, eps=1e-15)
    g = torch.Generator().manual_seed(3)
    N = 1000
    rows = (torch.randn(N, 3, generator=g), torch.randn(N, 3, 1, generator=g), torch.randn(N, 3, generator=g),
            torch.randn(N, 4, generator=g), torch.randn(N, 1, generator=g), torch.rand(N, 1, generator=g), torch.rand(N, 1, generator=g))
    n = extend_from_pcd(gm, *[t.to(dev) for t in rows])
    assert n == N and torch.equal(gm._xyz.detach().cpu(), rows[0]) and gm._features_rest.shape == (N, 0, 3)
    assert torch.equal(gm._features_dc.detach().cpu(), rows[1].transpose(1, 2).contiguous())
    assert gm.max_radii2D.shape == (N,) and float(gm.denom.sum()) == 0.0


def test_unkeyed_downsampling_draws_differ_per_call_and_agree_across_models(golden_dir):
    """create_pcd_from_image without kf_id / sample_idx / generator (the reference's own signature has no key-frame id,
    gaussian_model.py:118,170): successive calls on one model must draw DIFFERENT pixel subsets — the reference draws a fresh
    np.random.choice per call; round 4 keyed every such call by kf_id = -1 alike — and a second model (another replica) making
    the same calls in the same order must reproduce them."""
    from splatloc_amd import keyframe
    d = np.load(os.path.join(golden_dir, "keyframe.npz"))
    dev = torch.device(DEV)
    cam = _cam(d, dev)
    depth = torch.from_numpy(d["view_depth"]).to(dev)
    clouds = []
    for _model_id in range(2):
        gm = types.SimpleNamespace(max_sh_degree=0, isotropic=False,
                                   config={"Dataset": {"pcd_downsample": max(int(d["cfg"][0]), 4), "point_size": float(d["cfg"][1]),
                                                       "adaptive_pointsize": bool(d["cfg"][2])}})
        clouds.append([keyframe.create_pcd_from_image(gm, cam, depth)[0] for _ in range(3)])
    a, b = clouds
    assert not torch.equal(a[0], a[1]) and not torch.equal(a[1], a[2])
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    keyed = keyframe.create_pcd_from_image(types.SimpleNamespace(max_sh_degree=0, isotropic=False, config=gm.config), cam, depth,
                                           kf_id=7)[0]
    assert not torch.equal(keyed, a[0])
