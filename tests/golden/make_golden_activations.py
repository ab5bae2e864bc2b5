"""Generates tests/golden/activations.npz by running the reference's UNMODIFIED render()
(gaussian_splatting/gaussian_renderer/__init__.py:13-141) on seeded GaussianModels in THIS
container, with a recording stub in place of diff_gauss.GaussianRasterizer.

Captured per case (the activations + SH/feature packing that sit between the
raw optimiser parameters and the rasterizer call):
  raw parameters   _xyz, _features_dc, _features_rest, _scaling, _rotation, _opacity, _kp_score
                   (gaussian_model.py:40-55, 222-241) and the camera centre
  rasterizer args  scales = exp, rotations = normalize, opacities = sigmoid (gaussian_model.py:78-105),
                   colors_precomp = cat(clamp_min(eval_sh + 0.5, 0), kp_score)
                   (gaussian_renderer/__init__.py:84-102, sh_utils.py:55-118)
  gradients        of sum_k <arg_k, G_k> (seeded G_k) w.r.t. the raw parameters, by the
                   reference's own autograd graph
Cases: "deg0" = SplatLoc's configuration (max_sh_degree 0, f_rest [P,0,3]); "deg2of3" =
max_sh_degree 3 with active_sh_degree 2 (view-dependent colour: gradient reaches _xyz);
"deg3" = all 16 coefficients active.
Only the fixture (data) is committed; /root/reference never travels.
"""
import math
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402  (sets sys.path for the reference and our stubs)
import numpy as np  # noqa: E402
import torch  # noqa: E402


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
        import gaussian_splatting.gaussian_renderer as gr

        record = {}

        class Recorder(torch.nn.Module):
            def __init__(self, raster_settings):
                super().__init__()
                record["settings"] = raster_settings

            def forward(self, **kw):
                record["kwargs"] = kw
                P = kw["means3D"].shape[0]
                rs = record["settings"]
                C = kw["colors_precomp"].shape[1]
                z = lambda *s: torch.zeros(*s)  # noqa: E731
                return z(C, rs.image_height, rs.image_width), z(1, rs.image_height, rs.image_width), \
                    z(1, rs.image_height, rs.image_width), torch.zeros(P, dtype=torch.int32)

        gr.GaussianRasterizer = Recorder
        fx, fy, cx, cy, W, H = 320.0, 320.0, 319.5, 239.5, 640, 480
        proj = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=fx, fy=fy, cx=cx, cy=cy, W=W, H=H).transpose(0, 1)
        fovx, fovy = 2 * math.atan(W / (2 * fx)), 2 * math.atan(H / (2 * fy))
        opt = types.SimpleNamespace(percent_dense=0.01, position_lr_init=0.00016, position_lr_final=0.0000016,
                                    position_lr_delay_mult=0.01, position_lr_max_steps=30000, feature_lr=0.0025,
                                    opacity_lr=0.05, marker_lr=0.05, kp_score_lr=0.05, scaling_lr=0.001,
                                    rotation_lr=0.001)
        cfg = {"Training": {"primitive_reg": True}}
        for name, max_deg, active, P, seed in (("deg0", 0, 0, 600, 11), ("deg2of3", 3, 2, 400, 12),
                                               ("deg3", 3, 3, 300, 13)):
            g = torch.Generator().manual_seed(seed)
            T = torch.eye(4)
            T[:3, 3] = torch.tensor([0.2, -0.1, 0.4])
            cam = Camera(0, None, None, T, proj, fx, fy, cx, cy, fovx, fovy, H, W, None, None, device="cpu")
            gm = GaussianModel(max_deg, config=cfg)
            gm.init_lr(6.0)
            gm.training_setup(opt)
            K = (max_deg + 1) ** 2
            z = 0.8 + 4.0 * torch.rand(P, generator=g)
            xyz = torch.stack([(2 * torch.rand(P, generator=g) - 1) * z,
                               (2 * torch.rand(P, generator=g) - 1) * 0.75 * z, z], 1)
            feats = 0.8 * torch.randn(P, 3, K, generator=g)          # [P,3,K] as create_pcd_from_image builds it
            feats[:10, :, 0] = -3.0                                   # some colours clamp at 0
            gm.extend_from_pcd(xyz.clone(), feats, torch.log(0.03 * torch.exp(0.4 * torch.randn(P, 3, generator=g))),
                               torch.randn(P, 4, generator=g), 1.5 * torch.randn(P, 1, generator=g),
                               torch.rand(P, 1, generator=g), torch.rand(P, 1, generator=g))
            for _ in range(active):
                gm.oneupSHdegree()
            assert gm.active_sh_degree == active
            pipe = types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False)
            gr.render(cam, gm, pipe, torch.zeros(3))
            kw, rs = record["kwargs"], record["settings"]
            assert rs.sh_degree == active and kw["shs"] is None
            loss = 0
            G = {}
            for k in ("means3D", "colors_precomp", "opacities", "scales", "rotations"):
                G[k] = torch.randn(kw[k].shape, generator=g)
                loss = loss + (kw[k] * G[k]).sum()
            loss.backward()
            pre = name + "_"
            out[pre + "campos"] = rs.campos.detach().contiguous().numpy()
            out[pre + "active_sh_degree"] = np.array(active)
            out[pre + "max_sh_degree"] = np.array(max_deg)
            for k, t in (("xyz", gm._xyz), ("f_dc", gm._features_dc), ("f_rest", gm._features_rest),
                         ("scaling", gm._scaling), ("rotation", gm._rotation), ("opacity", gm._opacity),
                         ("kp_score", gm._kp_score)):
                out[pre + "raw_" + k] = t.detach().numpy().copy()
                out[pre + "grad_" + k] = (t.grad if t.grad is not None else torch.zeros_like(t)).numpy().copy()
            for k in G:
                out[pre + "out_" + k] = kw[k].detach().numpy().copy()
                out[pre + "G_" + k] = G[k].numpy().copy()
    path = os.path.join(HERE, "activations.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")
    for k in sorted(out):
        if k.startswith("deg2of3"):
            print("  ", k, out[k].shape, out[k].dtype)


if __name__ == "__main__":
    main()
