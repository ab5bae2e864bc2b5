"""Generates tests/golden/refine_step.npz: THREE consecutive iterations of SplatLoc.color_refinement
(train_gaussians.py:269-297 — 26 000 of the ~35 000 rasterizer calls of a scene) run by the reference's OWN Python in
THIS container:

    render_pkg = render(viewpoint_cam, gaussians, pipeline_params, background)     gaussian_renderer/__init__.py:13-141
    Ll1 = l1_loss(image, gt_image)                                                  loss_utils.py:21-22
    loss = (1 - lambda_dssim) * Ll1 + lambda_dssim * (1 - ssim(image, gt_image))    loss_utils.py:61-102, train_gaussians.py:285
    loss.backward()
    key_mask = get_marker.detach().squeeze() > 0.005;  get_xyz.grad[key_mask] = 0   train_gaussians.py:289-291 (primitive_reg)
    max_radii2D[visibility_filter] = max(max_radii2D[visibility_filter], radii[visibility_filter])     :294
    optimizer.step(); optimizer.zero_grad(set_to_none=True); update_learning_rate(iteration)           :295-297

on the reference's GaussianModel (extend_from_pcd + training_setup: torch.optim.Adam over the 8 groups, eps 1e-15,
lr schedule `helper`) and Camera objects.  The only piece that is not reference code is the rasterizer behind
`diff_gauss.GaussianRasterizer` (un-vendored CUDA): the CPU oracle stands in (make_golden_map_step.py).
The loop body is restated line by line because importing train_gaussians.py pulls GUI / OpenGL modules.

What the recording pins: only the RGB channels carry gradient (`kp_prob`, depth, opacity are not in the loss: the
kp_score column and `_marker` get NO gradient, hence no Adam state), the key-primitive gate on xyz, the max_radii2D
update (the reference indexes the float statistic with int32 radii), Adam's state carried from step to step and the
xyz learning rate after `update_learning_rate(iteration)`.  Only the fixture (data) is committed.
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

from make_golden_map_step import OracleRasterizer  # noqa: E402

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
        out[f"{pre}lr_{grp['name']}"] = np.array(grp["lr"])
    out[f"{pre}max_radii"] = gm.max_radii2D.numpy().copy()


def main():
    for m in ("cv2", "open3d", "tinycudann", "models"):
        mg.stub(m)
    mg.stub("plyfile", PlyData=object, PlyElement=object)
    mg.stub("models.decoders", FeatureDecoder=object)
    out = {}
    with mg.CudaToCpu():
        from gaussian_splatting.utils.graphics_utils import getProjectionMatrix2
        from gaussian_splatting.utils.loss_utils import l1_loss, ssim
        from gaussian_splatting.scene.gaussian_model import GaussianModel
        from utils.camera_utils import Camera
        import gaussian_splatting.gaussian_renderer as gr
        gr.GaussianRasterizer = OracleRasterizer
        g = torch.Generator().manual_seed(4242)
        W, H = 96, 72
        fx = fy = 48.0
        cx, cy = 47.5, 35.5
        proj = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=fx, fy=fy, cx=cx, cy=cy, W=W, H=H).transpose(0, 1)
        fovx, fovy = 2 * math.atan(W / (2 * fx)), 2 * math.atan(H / (2 * fy))
        config = {"Training": {"primitive_reg": True, "rgb_boundary_threshold": 0.01}}
        opt = types.SimpleNamespace(percent_dense=0.01, position_lr_init=0.0016, position_lr_final=0.0000016,
                                    position_lr_delay_mult=0.01, position_lr_max_steps=30000, feature_lr=0.0025,
                                    opacity_lr=0.05, marker_lr=0.05, kp_score_lr=0.05, scaling_lr=0.001,
                                    rotation_lr=0.001, lambda_dssim=0.2)      # configs/replica_nerf/base_config.yaml:66-80
        P = 2500
        gaussians = GaussianModel(0, config=config)
        gaussians.init_lr(6.0)
        gaussians.training_setup(opt)
        z = 0.8 + 4.0 * torch.rand(P, generator=g)
        xyz = torch.stack([(2 * torch.rand(P, generator=g) - 1) * z, (2 * torch.rand(P, generator=g) - 1) * 0.75 * z, z], 1)
        feats = 0.8 * torch.randn(P, 3, 1, generator=g)
        markers = (torch.rand(P, 1, generator=g) < 0.3).float() * torch.rand(P, 1, generator=g) * 0.9
        gaussians.extend_from_pcd(xyz.clone(), feats, torch.log(0.06 * torch.exp(0.4 * torch.randn(P, 3, generator=g))),
                                  torch.randn(P, 4, generator=g), 1.5 * torch.randn(P, 1, generator=g), markers,
                                  torch.randn(P, 1, generator=g))
        gaussians.max_radii2D = torch.randint(0, 12, (P,), generator=g).float()   # a statistic already in progress
        viewpoints = []
        for k in range(3):
            T = torch.eye(4)
            ang = 0.06 * (k - 1)
            T[:3, :3] = torch.tensor([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
            T[:3, 3] = torch.tensor([0.04 * k, -0.03 * k, 0.1])
            color = torch.rand(3, H, W, generator=g)
            depth = (0.5 + 3 * torch.rand(H, W, generator=g)).numpy()
            kp = torch.rand(H, W, generator=g) ** 4
            cam = Camera(k, color, depth, T, proj, fx, fy, cx, cy, fovx, fovy, H, W, kp, None, device="cpu")
            viewpoints.append(cam)
            out[f"view{k}_T"] = T.numpy().copy()
            out[f"view{k}_color"] = color.numpy().copy()
        out["intr"] = np.array([fx, fy, cx, cy, W, H, math.tan(fovx * 0.5), math.tan(fovy * 0.5)])
        out["lambda_dssim"] = np.array(opt.lambda_dssim)
        snapshot(gaussians, "s0_", out)
        pipeline_params = types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False)
        background = torch.tensor([0, 0, 0], dtype=torch.float32)          # train_gaussians.py:70
        primitive_reg = config["Training"]["primitive_reg"]

        # ---- SplatLoc.color_refinement loop body, train_gaussians.py:272-297, line by line ----
        for iteration in range(1, 4):
            viewpoint_cam = viewpoints[iteration - 1]
            render_pkg = gr.render(viewpoint_cam, gaussians, pipeline_params, background)
            image, visibility_filter, radii = (render_pkg["render"], render_pkg["visibility_filter"], render_pkg["radii"])
            gt_image = viewpoint_cam.original_image.cuda()
            Ll1 = l1_loss(image, gt_image)
            loss = (1.0 - opt.lambda_dssim) * (Ll1) + opt.lambda_dssim * (1.0 - ssim(image, gt_image))
            loss.backward()
            if primitive_reg:
                key_mask = gaussians.get_marker.detach().squeeze() > 0.005
                gaussians.get_xyz.grad[key_mask] = 0
            pre = f"it{iteration}_"
            out[pre + "loss"] = np.array(loss.item())
            out[pre + "l1"] = np.array(Ll1.item())
            out[pre + "radii"] = radii.numpy().copy()
            for name in GROUPS:
                gr_ = getattr(gaussians, ATTR[name]).grad
                out[pre + "has_grad_" + name] = np.array(gr_ is not None)
                if gr_ is not None and gr_.numel():
                    out[pre + "grad_" + name] = gr_.numpy().copy()       # xyz: AFTER the key-primitive gate
            with torch.no_grad():
                gaussians.max_radii2D[visibility_filter] = torch.max(gaussians.max_radii2D[visibility_filter],
                                                                     radii[visibility_filter])
                gaussians.optimizer.step()
                gaussians.optimizer.zero_grad(set_to_none=True)
                gaussians.update_learning_rate(iteration)
            snapshot(gaussians, f"s{iteration}_", out)
            print("iteration", iteration, "loss", loss.item(), "visible", int(visibility_filter.sum()),
                  "kp grad", out[pre + "has_grad_kp_score"], "marker grad", out[pre + "has_grad_marker"])
    path = os.path.join(HERE, "refine_step.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
