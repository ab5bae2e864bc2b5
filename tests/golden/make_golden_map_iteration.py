"""Generates tests/golden/map_iteration.npz: THREE complete iterations of the loop body of SplatLoc.map
(train_gaussians.py:188-267) run by the reference's OWN Python in THIS container — render x 5 views, get_loss_mapping +
get_loss_marker, the isotropic regulariser, backward, the key-primitive gate, max_radii2D / add_densification_stats per
view, reset_opacity_nonvisible (iteration 2: `iteration_count % gaussian_reset == 0`), optimizer.step / zero_grad /
update_learning_rate — on the reference's GaussianModel (torch.optim.Adam over the 8 groups) and Camera objects, with the
CPU oracle standing in for the un-vendored rasterizer (make_golden_map_step.py).  The loop body is restated line by line
because importing train_gaussians.py pulls GUI / OpenGL modules.  Recorded after every iteration: the 8 parameter tensors,
Adam moments / step counters / learning rates, xyz_gradient_accum, denom, max_radii2D, the loss.  (densify_and_prune is
pinned separately by densify.npz; `gaussian_update_every` is set out of reach here.)  Only the fixture is committed.
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
    out[f"{pre}accum"] = gm.xyz_gradient_accum.numpy().copy()
    out[f"{pre}denom"] = gm.denom.numpy().copy()
    out[f"{pre}max_radii"] = gm.max_radii2D.numpy().copy()


def main():
    for m in ("cv2", "open3d", "tinycudann", "models"):
        mg.stub(m)
    mg.stub("plyfile", PlyData=object, PlyElement=object)
    mg.stub("models.decoders", FeatureDecoder=object)
    out = {}
    with mg.CudaToCpu():
        from gaussian_splatting.utils.graphics_utils import getProjectionMatrix2
        from gaussian_splatting.scene.gaussian_model import GaussianModel
        from utils.camera_utils import Camera
        from utils.utils import get_loss_mapping
        import gaussian_splatting.gaussian_renderer as gr
        gr.GaussianRasterizer = OracleRasterizer
        g = torch.Generator().manual_seed(31337)
        W, H = 96, 72
        fx = fy = 48.0
        cx, cy = 47.5, 35.5
        proj = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=fx, fy=fy, cx=cx, cy=cy, W=W, H=H).transpose(0, 1)
        fovx, fovy = 2 * math.atan(W / (2 * fx)), 2 * math.atan(H / (2 * fy))
        config = {"Training": {"primitive_reg": True, "rgb_boundary_threshold": 0.01}}
        opt = types.SimpleNamespace(percent_dense=0.01, position_lr_init=0.0016, position_lr_final=0.0000016,
                                    position_lr_delay_mult=0.01, position_lr_max_steps=30000, feature_lr=0.0025,
                                    opacity_lr=0.05, marker_lr=0.05, kp_score_lr=0.05, scaling_lr=0.001,
                                    rotation_lr=0.001)
        P = 2000
        gaussians = GaussianModel(0, config=config)
        gaussians.init_lr(6.0)
        gaussians.training_setup(opt)
        z = 0.8 + 4.0 * torch.rand(P, generator=g)
        xyz = torch.stack([(2 * torch.rand(P, generator=g) - 1) * z, (2 * torch.rand(P, generator=g) - 1) * 0.75 * z, z], 1)
        markers = (torch.rand(P, 1, generator=g) < 0.3).float() * torch.rand(P, 1, generator=g) * 0.9
        gaussians.extend_from_pcd(xyz.clone(), 0.8 * torch.randn(P, 3, 1, generator=g),
                                  torch.log(0.06 * torch.exp(0.4 * torch.randn(P, 3, generator=g))), torch.randn(P, 4, generator=g),
                                  1.5 * torch.randn(P, 1, generator=g), markers, torch.randn(P, 1, generator=g))
        viewpoints = []
        for k in range(7):
            T = torch.eye(4)
            ang = 0.05 * (k - 3)
            T[:3, :3] = torch.tensor([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
            T[:3, 3] = torch.tensor([0.05 * k, -0.02 * k, 0.1])
            color = torch.rand(3, H, W, generator=g)
            color[:, :3] = 0.0
            depth = (0.5 + 3 * torch.rand(H, W, generator=g)).numpy()
            depth[:, :5] = 0.0
            kp = torch.rand(H, W, generator=g) ** 4
            cam = Camera(k, color, depth, T, proj, fx, fy, cx, cy, fovx, fovy, H, W, kp, None, device="cpu")
            viewpoints.append(cam)
            out[f"view{k}_T"], out[f"view{k}_color"] = T.numpy().copy(), color.numpy().copy()
            out[f"view{k}_depth"], out[f"view{k}_kp"] = depth.copy(), kp.numpy().copy()
        out["intr"] = np.array([fx, fy, cx, cy, W, H])
        snapshot(gaussians, "s0_", out)
        pipeline_params = types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False)
        background = torch.tensor([0, 0, 0], dtype=torch.float32)
        primitive_reg = True
        gaussian_update_every, gaussian_update_offset, gaussian_reset = 150, 50, 2
        out["gaussian_reset"] = np.array(gaussian_reset)
        iteration_count = 0
        for it in range(3):
            # ---- SplatLoc.map loop body, train_gaussians.py:188-267, line by line ----
            iteration_count += 1
            loss_mapping = 0
            viewspace_point_tensor_acm, visibility_filter_acm, radii_acm = [], [], []
            perm = torch.randperm(len(viewpoints), generator=g)[:5]
            out[f"it{iteration_count}_views"] = perm.numpy().copy()
            for cam_idx in perm:
                viewpoint = viewpoints[cam_idx]
                render_pkg = gr.render(viewpoint, gaussians, pipeline_params, background)
                image, marker, viewspace_point_tensor, visibility_filter, radii, depth, opacity = (
                    render_pkg["render"], render_pkg["kp_prob"], render_pkg["viewspace_points"],
                    render_pkg["visibility_filter"], render_pkg["radii"], render_pkg["depth"], render_pkg["opacity"])
                loss_mapping += get_loss_mapping(config, image, depth, viewpoint, opacity)
                pred = torch.sigmoid(marker.view(-1))
                loss_mapping += torch.nn.functional.binary_cross_entropy(pred, viewpoint.kp_score.view(-1).float(), reduction="mean")
                viewspace_point_tensor_acm.append(viewspace_point_tensor)
                visibility_filter_acm.append(visibility_filter)
                radii_acm.append(radii)
            scaling = gaussians.get_scaling
            score = gaussians.get_marker.detach()
            mask = score.cpu().squeeze() > 0.005
            isotropic_loss = torch.abs(scaling.mean(dim=1).view(-1, 1)[mask] / (0.02 * (1 - score[mask])) - 1)
            if primitive_reg:
                loss_mapping += 0.01 * isotropic_loss.mean()
            loss_mapping.backward()
            if primitive_reg:
                key_mask = gaussians.get_marker.detach().cpu().squeeze() > 0.005
                gaussians.get_xyz.grad[key_mask] = 0
            with torch.no_grad():
                for idx in range(len(viewspace_point_tensor_acm)):
                    gaussians.max_radii2D[visibility_filter_acm[idx]] = torch.max(
                        gaussians.max_radii2D[visibility_filter_acm[idx]], radii_acm[idx][visibility_filter_acm[idx]])
                    gaussians.add_densification_stats(viewspace_point_tensor_acm[idx], visibility_filter_acm[idx])
                update_gaussian = (iteration_count % gaussian_update_every == gaussian_update_offset)
                assert not update_gaussian
                if (iteration_count % gaussian_reset) == 0 and (not update_gaussian):
                    gaussians.reset_opacity_nonvisible(visibility_filter_acm)
                gaussians.optimizer.step()
                gaussians.optimizer.zero_grad(set_to_none=True)
                gaussians.update_learning_rate(iteration_count)
            out[f"it{iteration_count}_loss"] = np.array(loss_mapping.item())
            snapshot(gaussians, f"s{iteration_count}_", out)
            print("iteration", iteration_count, "loss", loss_mapping.item(), "views", perm.tolist(),
                  "reset" if iteration_count % gaussian_reset == 0 else "")
    path = os.path.join(HERE, "map_iteration.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
