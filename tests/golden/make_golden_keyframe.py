"""Generates tests/golden/keyframe.npz: ONE key-frame insertion of SplatLoc (train_gaussians.py:173-177, :332 ->
GaussianModel.extend_from_pcd_seq, gaussian_model.py:243-248) run by the reference's OWN Python in THIS container:

    create_pcd_from_image              gaussian_model.py:118-131   exposure affine, clamp, uint8 colours
    create_pcd_from_image_and_depth_score   :170-217  key / non-key pixel masks (score > 0.005), unprojection with the
                                       camera-to-world pose (creat_pcsd_from_mask :133-168), np.random.choice
                                       down-sampling of the non-key pixels, RGB2SH, the ONLY caller of
                                       simple_knn._C.distCUDA2 (:206):
                                           dist2  = clamp_min(distCUDA2(xyz), 1e-7) * point_size
                                           scales = log(sqrt(dist2))[..., None].repeat(1, 3)
                                       unit quaternions, opacity logit(0.5), markers = SuperPoint scores, kp_score 0.5
    extend_from_pcd                    :222-241  -> densification_postfix / cat_tensors_to_optimizer :528-587:
                                       rows appended to the 8 parameter tensors, ZERO Adam moments appended to the
                                       groups that have state, statistics reset

on a GaussianModel that already holds 700 Gaussians with live Adam state (three seeded steps; `_marker` never gets a
gradient, hence no state).  Not reference code: `simple_knn._C.distCUDA2` (un-vendored CUDA) is stood in by
the CPU oracle's exact 3-NN (oracle.dist2), and `np.random.choice` is replaced by a recorded draw so that the device
path can be given the same sample.  Only the fixture (data) is committed.
"""
import math
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import oracle  # noqa: E402

GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "marker", "kp_score", "scaling", "rotation")
ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
        "marker": "_marker", "kp_score": "_kp_score", "scaling": "_scaling", "rotation": "_rotation"}


def snapshot(gm, pre, out):
    for name in GROUPS:
        out[f"{pre}{name}"] = getattr(gm, ATTR[name]).detach().numpy().copy()
    for grp in gm.optimizer.param_groups:
        st = gm.optimizer.state.get(grp["params"][0], None)
        out[f"{pre}has_state_{grp['name']}"] = np.array(bool(st is not None and len(st)))
        if st is not None and len(st):
            out[f"{pre}m_{grp['name']}"] = st["exp_avg"].numpy().copy()
            out[f"{pre}v_{grp['name']}"] = st["exp_avg_sq"].numpy().copy()
            out[f"{pre}step_{grp['name']}"] = np.array(float(st["step"]))
    out[f"{pre}accum"] = gm.xyz_gradient_accum.numpy().copy()
    out[f"{pre}denom"] = gm.denom.numpy().copy()
    out[f"{pre}max_radii"] = gm.max_radii2D.numpy().copy()


def main():
    for m in ("cv2", "open3d", "tinycudann", "models"):
        mg.stub(m)
    mg.stub("plyfile", PlyData=object, PlyElement=object)
    mg.stub("models.decoders", FeatureDecoder=object)
    knn_calls = []

    def dist_stub(points):       # stands in for the un-vendored CUDA kernel: exact 3-NN mean squared distance
        knn_calls.append(points.detach().numpy().copy())
        return torch.from_numpy(oracle.dist2(points.detach().numpy()))

    mg.stub("simple_knn")
    mg.stub("simple_knn._C", distCUDA2=dist_stub)
    out = {}
    with mg.CudaToCpu():
        from gaussian_splatting.utils.graphics_utils import getProjectionMatrix2
        from gaussian_splatting.scene.gaussian_model import GaussianModel
        import gaussian_splatting.scene.gaussian_model as gmod
        from utils.camera_utils import Camera
        g = torch.Generator().manual_seed(777)
        W, H = 160, 120
        fx = fy = 80.0
        cx, cy = 79.5, 59.5
        proj = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=fx, fy=fy, cx=cx, cy=cy, W=W, H=H).transpose(0, 1)
        fovx, fovy = 2 * math.atan(W / (2 * fx)), 2 * math.atan(H / (2 * fy))
        config = {"Training": {"primitive_reg": True, "rgb_boundary_threshold": 0.01},
                  "Dataset": {"pcd_downsample": 64, "point_size": 0.01, "adaptive_pointsize": True}}   # configs/replica_nerf/base_config.yaml
        opt = types.SimpleNamespace(percent_dense=0.01, position_lr_init=0.0016, position_lr_final=0.0000016,
                                    position_lr_delay_mult=0.01, position_lr_max_steps=30000, feature_lr=0.0025,
                                    opacity_lr=0.05, marker_lr=0.05, kp_score_lr=0.05, scaling_lr=0.001,
                                    rotation_lr=0.001)
        gm = GaussianModel(0, config=config)
        gm.init_lr(6.0)
        gm.training_setup(opt)
        P0 = 700
        z = 0.8 + 4.0 * torch.rand(P0, generator=g)
        xyz = torch.stack([(2 * torch.rand(P0, generator=g) - 1) * z, (2 * torch.rand(P0, generator=g) - 1) * 0.75 * z, z], 1)
        gm.extend_from_pcd(xyz.clone(), 0.8 * torch.randn(P0, 3, 1, generator=g),
                           torch.log(0.06 * torch.exp(0.4 * torch.randn(P0, 3, generator=g))), torch.randn(P0, 4, generator=g),
                           1.5 * torch.randn(P0, 1, generator=g), torch.rand(P0, 1, generator=g) * 0.01,
                           torch.randn(P0, 1, generator=g))
        for k in range(3):      # live Adam state (the marker never gets a gradient in SplatLoc)
            for name in GROUPS:
                p = getattr(gm, ATTR[name])
                p.grad = None if name == "marker" else torch.randn(p.shape, generator=g) * 1e-3
            gm.optimizer.step()
            gm.optimizer.zero_grad(set_to_none=True)
        gm.xyz_gradient_accum += 0.5        # statistics in progress: the insertion resets them
        gm.denom += 2.0
        gm.max_radii2D += 7.0
        snapshot(gm, "before_", out)

        # the new key-frame
        T = torch.eye(4)
        ang = 0.2
        T[:3, :3] = torch.tensor([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
        T[:3, 3] = torch.tensor([0.3, -0.1, 0.2])
        color = torch.rand(3, H, W, generator=g)
        depth = (0.6 + 3 * torch.rand(H, W, generator=g)).numpy().astype(np.float32)
        depth[:, :7] = 0.0                                      # invalid depth: no point
        kp = torch.rand(H, W, generator=g) ** 6                 # sparse SuperPoint-like scores: ~40 % above 0.005
        kp[::2] *= 0.001
        cam = Camera(0, color, depth, T, proj, fx, fy, cx, cy, fovx, fovy, H, W, kp, None, device="cpu")
        with torch.no_grad():
            cam.exposure_a.fill_(0.05)
            cam.exposure_b.fill_(-0.02)
        out["view_T"], out["view_color"], out["view_depth"], out["view_kp"] = T.numpy().copy(), color.numpy().copy(), depth.copy(), kp.numpy().copy()
        out["view_exposure"] = np.array([0.05, -0.02], np.float32)
        out["intr"] = np.array([fx, fy, cx, cy, W, H])
        out["cfg"] = np.array([64, 0.01, 1.0])                  # pcd_downsample, point_size, adaptive_pointsize

        # np.random.choice -> a recorded draw (with replacement, like np.random.choice's default)
        rng = np.random.default_rng(99)
        draws = []

        def choice(n_points, n_samples):
            idx = rng.integers(0, n_points, size=n_samples)
            draws.append(idx.copy())
            return idx

        real_choice = np.random.choice
        np.random.choice = choice
        try:
            tensors = gm.create_pcd_from_image(cam, depthmap=depth)
        finally:
            np.random.choice = real_choice
        assert len(draws) == 1 and len(knn_calls) == 1
        out["sample_idx"] = draws[0]
        names = ("fused_point_cloud", "features", "scales", "rots", "opacities", "markers", "kp_scores")
        for n, t in zip(names, tensors):
            out["pcd_" + n] = t.detach().numpy().copy()
        out["knn_points"] = knn_calls[0]
        out["knn_dist2"] = oracle.dist2(knn_calls[0])
        n_kp = int(((depth > 0) & (kp.numpy() > 0.005)).sum())
        out["num_kp"] = np.array(n_kp)
        gm.extend_from_pcd(*tensors)
        snapshot(gm, "after_", out)
        print("key pixels", n_kp, "sampled non-key", draws[0].shape[0], "new rows", tensors[0].shape[0], "model rows",
              gm._xyz.shape[0], "scale range", float(tensors[2].min()), float(tensors[2].max()))
        del gmod
    path = os.path.join(HERE, "keyframe.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
