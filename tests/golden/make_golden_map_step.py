"""Generates tests/golden/map_step.npz: ONE map()-shaped optimisation step of SplatLoc
run by the reference's OWN Python in THIS container —

    for 5 views:  render(viewpoint, gaussians, pipeline_params, background)      gaussian_renderer/__init__.py:13-141
                  loss += get_loss_mapping(config, image, depth, viewpoint, opacity)      utils/utils.py:55-82
                  loss += get_loss_marker(config, marker, viewpoint.kp_score)             train_gaussians.py:38-42
    loss += 0.01 * isotropic_loss.mean()                                         train_gaussians.py:221-228
    loss.backward()                                                              train_gaussians.py:229

— on the reference's GaussianModel (extend_from_pcd + training_setup) and Camera objects.  The only piece
that is not reference code is the rasterizer behind `diff_gauss.GaussianRasterizer` (un-vendored CUDA,
SURVEY F1): an autograd.Function backed by the CPU oracle stands in for it.  The loop body of
SplatLoc.map is restated line by line below because importing train_gaussians.py pulls GUI / OpenGL
modules.  Recorded: every input, the loss, and the gradient of each of the 8 parameter tensors
(`_marker.grad is None`, `_features_rest` is [P,0,3]), each view's screen-space gradient and radii, and the
exposure gradients.  Only the fixture (data) is committed.
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


class _OracleRaster(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, colors, opacities, scales, rotations, rs):
        st = oracle.Settings(rs.image_height, rs.image_width, rs.tanfovx, rs.tanfovy, rs.scale_modifier, rs.sh_degree)
        f = oracle.forward(st, rs.bg.numpy(), means3D.detach().numpy(), opacities.detach().numpy(),
                           rs.viewmatrix.numpy(), rs.projmatrix.numpy(), rs.campos.contiguous().numpy(),
                           colors_precomp=colors.detach().numpy(), scales=scales.detach().numpy(),
                           rotations=rotations.detach().numpy(), omp=True)
        ctx.f = f
        radii = torch.from_numpy(f["radii"].copy())
        ctx.mark_non_differentiable(radii)
        return torch.from_numpy(f["color"].copy()), torch.from_numpy(f["depth"].copy()), \
            torch.from_numpy(f["alpha"].copy()), radii

    @staticmethod
    def backward(ctx, g_color, g_depth, g_alpha, _g_radii):
        z = lambda g, ref: np.zeros_like(ref) if g is None else g.numpy()  # noqa: E731
        f = ctx.f
        b = oracle.backward(f, z(g_color, f["color"]), z(g_depth, f["depth"]), z(g_alpha, f["alpha"]), omp=True)
        t = torch.from_numpy
        return (t(b["dL_dmeans3D"]), t(b["dL_dmeans2D"]), t(b["dL_dcolors"]), t(b["dL_dopacities"]), t(b["dL_dscales"]),
                t(b["dL_drotations"]), None)


class OracleRasterizer(torch.nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.rs = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        assert shs is None and cov3D_precomp is None
        return _OracleRaster.apply(means3D, means2D, colors_precomp, opacities, scales, rotations, self.rs)


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
        g = torch.Generator().manual_seed(2024)
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
        assert gaussians._features_rest.shape == (P, 0, 3)
        viewpoints = []
        for k in range(5):
            T = torch.eye(4)
            ang = 0.05 * (k - 2)
            T[:3, :3] = torch.tensor([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
            T[:3, 3] = torch.tensor([0.05 * k, -0.02 * k, 0.1])
            color = torch.rand(3, H, W, generator=g)
            color[:, :3] = 0.0                                             # below the rgb boundary threshold
            depth = (0.5 + 3 * torch.rand(H, W, generator=g)).numpy()
            depth[:, :5] = 0.0                                             # invalid depth
            kp = torch.rand(H, W, generator=g) ** 4                        # float score map (soft BCE targets)
            cam = Camera(k, color, depth, T, proj, fx, fy, cx, cy, fovx, fovy, H, W, kp, None, device="cpu")
            with torch.no_grad():
                cam.exposure_a.fill_(0.03 * (k - 2))
                cam.exposure_b.fill_(-0.01 * k)
            viewpoints.append(cam)
            out[f"view{k}_T"] = T.numpy().copy()
            out[f"view{k}_color"] = color.numpy().copy()
            out[f"view{k}_depth"] = depth.copy()
            out[f"view{k}_kp"] = kp.numpy().copy()
            out[f"view{k}_exposure"] = np.array([cam.exposure_a.item(), cam.exposure_b.item()], np.float32)
        out["intr"] = np.array([fx, fy, cx, cy, W, H, math.tan(fovx * 0.5), math.tan(fovy * 0.5)])
        names = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
                 "marker": "_marker", "kp_score": "_kp_score", "scaling": "_scaling", "rotation": "_rotation"}
        for k, a in names.items():
            out["raw_" + k] = getattr(gaussians, a).detach().numpy().copy()
        pipeline_params = types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False)
        background = torch.tensor([0, 0, 0], dtype=torch.float32)          # train_gaussians.py:70

        # ---- SplatLoc.map loop body, train_gaussians.py:190-229, line by line ----
        loss_mapping = 0
        viewspace_point_tensor_acm, visibility_filter_acm, radii_acm = [], [], []
        for viewpoint in viewpoints:
            render_pkg = gr.render(viewpoint, gaussians, pipeline_params, background)
            image, marker, viewspace_point_tensor, visibility_filter, radii, depth, opacity = (
                render_pkg["render"], render_pkg["kp_prob"], render_pkg["viewspace_points"],
                render_pkg["visibility_filter"], render_pkg["radii"], render_pkg["depth"], render_pkg["opacity"])
            loss_mapping += get_loss_mapping(config, image, depth, viewpoint, opacity)
            pred = torch.sigmoid(marker.view(-1))                                            # get_loss_marker
            loss_mapping += torch.nn.functional.binary_cross_entropy(pred, viewpoint.kp_score.view(-1).float(),
                                                                     reduction="mean")
            viewspace_point_tensor_acm.append(viewspace_point_tensor)
            visibility_filter_acm.append(visibility_filter)
            radii_acm.append(radii)
        scaling = gaussians.get_scaling
        score = gaussians.get_marker.detach()
        mask = score.cpu().squeeze() > 0.005
        isotropic_loss = torch.abs(scaling.mean(dim=1).view(-1, 1)[mask] / (0.02 * (1 - score[mask])) - 1)
        loss_mapping += 0.01 * isotropic_loss.mean()
        loss_mapping.backward()

        out["loss"] = np.array(loss_mapping.item())
        assert gaussians._marker.grad is None
        for k, a in names.items():
            gr_ = getattr(gaussians, a).grad
            out["has_grad_" + k] = np.array(gr_ is not None)
            if gr_ is not None:
                out["grad_" + k] = gr_.numpy().copy()
        for k in range(5):
            out[f"view{k}_viewspace_grad"] = viewspace_point_tensor_acm[k].grad.numpy().copy()
            out[f"view{k}_radii"] = radii_acm[k].numpy().copy()
            out[f"view{k}_exposure_grad"] = np.array([viewpoints[k].exposure_a.grad.item(), viewpoints[k].exposure_b.grad.item()])
        print("loss", out["loss"], "visible per view", [int((r > 0).sum()) for r in radii_acm])
    path = os.path.join(HERE, "map_step.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
