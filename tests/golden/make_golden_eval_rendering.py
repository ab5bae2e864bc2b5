"""Generates tests/golden/eval_rendering.npz: the per-frame loop body of the reference's `eval_rendering`
(utils/eval_utils.py:22-72) run by the reference's OWN Python in THIS container on a seeded model and three cameras at
the 12-Scenes intrinsics (configs/scenes12/base_config.yaml:17-27: fx = fy = 572, cx = 320, cy = 240 at 640x480, scaled
to 160x120):

    render_results = render(frame, gaussians, pipe, background)              gaussian_renderer/__init__.py:13-141
    image = torch.clamp(render_results["render"], 0.0, 1.0)
    mask = gt_image.cpu() > 0
    psnr_score = psnr(image[mask].unsqueeze(0), gt_image[mask].unsqueeze(0))   gaussian_splatting/utils/image_utils.py:19-21
    ssim_score = ssim(image.unsqueeze(0), gt_image.unsqueeze(0))               gaussian_splatting/utils/loss_utils.py:61-102

`utils/eval_utils.py` itself cannot be imported here (evo, open3d, torchmetrics: absent), so the loop body is restated
line by line around the imported `render`, `psnr`, `ssim`; LPIPS needs the torchmetrics AlexNet and is left out.  The
un-vendored rasterizer behind `diff_gauss.GaussianRasterizer` is the CPU oracle (make_golden_map_step.py).  The ground
truth of a frame is the render of a PERTURBED copy of the model (so PSNR is finite and frame dependent) with a band of
exactly-zero pixels (the mask) and values above 1 in the render (the clamp).  Only the fixture (data) is committed.
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


def main():
    for m in ("cv2", "open3d", "tinycudann", "models"):
        mg.stub(m)
    mg.stub("plyfile", PlyData=object, PlyElement=object)
    mg.stub("models.decoders", FeatureDecoder=object)
    out = {}
    with mg.CudaToCpu():
        from gaussian_splatting.utils.graphics_utils import getProjectionMatrix2
        from gaussian_splatting.utils.image_utils import psnr
        from gaussian_splatting.utils.loss_utils import ssim
        from gaussian_splatting.scene.gaussian_model import GaussianModel
        from utils.camera_utils import Camera
        import gaussian_splatting.gaussian_renderer as gr
        gr.GaussianRasterizer = OracleRasterizer
        g = torch.Generator().manual_seed(777)
        W, H = 160, 120
        s = W / 640.0
        fx = fy = 572.0 * s
        cx, cy = 320.0 * s, 240.0 * s
        proj = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=fx, fy=fy, cx=cx, cy=cy, W=W, H=H).transpose(0, 1)
        fovx, fovy = 2 * math.atan(W / (2 * fx)), 2 * math.atan(H / (2 * fy))
        config = {"Training": {"primitive_reg": True, "rgb_boundary_threshold": 0.01}}
        opt = types.SimpleNamespace(percent_dense=0.01, position_lr_init=0.0016, position_lr_final=0.0000016,
                                    position_lr_delay_mult=0.01, position_lr_max_steps=30000, feature_lr=0.0025,
                                    opacity_lr=0.05, marker_lr=0.05, kp_score_lr=0.05, scaling_lr=0.001,
                                    rotation_lr=0.001, lambda_dssim=0.2)
        P = 4000

        def model(noise):
            gg = torch.Generator().manual_seed(31)
            gm = GaussianModel(0, config=config)
            gm.init_lr(6.0)
            gm.training_setup(opt)
            z = 0.8 + 3.0 * torch.rand(P, generator=gg)
            xyz = torch.stack([(2 * torch.rand(P, generator=gg) - 1) * 0.6 * z, (2 * torch.rand(P, generator=gg) - 1) * 0.45 * z, z], 1)
            feats = 1.6 * torch.randn(P, 3, 1, generator=gg)          # rgb = 0.28 f + 0.5: some colours leave [0, 1]
            scal = torch.log(0.05 * torch.exp(0.4 * torch.randn(P, 3, generator=gg)))
            rot = torch.randn(P, 4, generator=gg)
            opa = 1.5 * torch.randn(P, 1, generator=gg) + 1.0
            if noise:
                n = torch.Generator().manual_seed(32)
                feats = feats + noise * torch.randn(P, 3, 1, generator=n)
                xyz = xyz + 0.3 * noise * 0.05 * torch.randn(P, 3, generator=n)
            gm.extend_from_pcd(xyz, feats, scal, rot, opa, torch.zeros(P, 1), torch.randn(P, 1, generator=gg))
            return gm

        gaussians, gt_model = model(0.0), model(0.35)
        pipe = types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False)
        background = torch.tensor([0, 0, 0], dtype=torch.float32)
        for name, gm in (("model", gaussians),):
            for k, a in (("xyz", "_xyz"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("opacity", "_opacity"),
                         ("kp_score", "_kp_score"), ("scaling", "_scaling"), ("rotation", "_rotation")):
                out[f"{name}_{k}"] = getattr(gm, a).detach().numpy().copy()
        psnr_array, ssim_array = [], []
        for k in range(3):
            T = torch.eye(4)
            ang = 0.08 * (k - 1)
            T[:3, :3] = torch.tensor([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
            T[:3, 3] = torch.tensor([0.05 * k, -0.02 * k, 0.1 * k])
            frame = Camera(k, torch.zeros(3, H, W), np.zeros((H, W), np.float32), T, proj, fx, fy, cx, cy, fovx, fovy, H, W,
                           torch.zeros(H, W), None, device="cpu")
            with torch.no_grad():
                gt_image = torch.clamp(gr.render(frame, gt_model, pipe, background)["render"], 0.0, 1.0).clone()
                gt_image[:, : 10 + 7 * k, :] = 0.0          # invalid band: exactly zero -> outside the PSNR mask
                gt_image[1, :, -(5 + k):] = 0.0             # one channel only: the mask is per ELEMENT, not per pixel
                # ---- utils/eval_utils.py:44-52, line by line ----
                render_results = gr.render(frame, gaussians, pipe, background)
                rendering = render_results["render"]
                image = torch.clamp(rendering, 0.0, 1.0)
                mask = gt_image.cpu() > 0
                psnr_score = psnr((image[mask]).unsqueeze(0), (gt_image[mask]).unsqueeze(0))
                ssim_score = ssim((image).unsqueeze(0), (gt_image).unsqueeze(0))
            psnr_array.append(psnr_score.item())
            ssim_array.append(ssim_score.item())
            out[f"view{k}_T"] = T.numpy().copy()
            out[f"view{k}_gt"] = gt_image.numpy().copy()
            out[f"view{k}_render"] = rendering.numpy().copy()          # un-clamped, as the rasterizer returns it
            out[f"view{k}_mask_count"] = np.array(int(mask.sum()))
        out["intr"] = np.array([fx, fy, cx, cy, W, H, math.tan(fovx * 0.5), math.tan(fovy * 0.5)])
        out["psnr"] = np.array(psnr_array)
        out["ssim"] = np.array(ssim_array)
        out["mean_psnr"] = np.array(float(np.mean(psnr_array)))       # eval_utils.py:59-60
        out["mean_ssim"] = np.array(float(np.mean(ssim_array)))
    path = os.path.join(HERE, "eval_rendering.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), out["psnr"], out["ssim"],
          [float((out[f"view{k}_render"] > 1).mean()) for k in range(3)])


if __name__ == "__main__":
    main()
