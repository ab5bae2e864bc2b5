"""Generates tests/golden/*.npz by importing the reference's own Python in THIS container.

Run once here (`python tests/golden/make_golden.py`); /root/reference does not exist on
the GPU box, so only the small fixtures (data: inputs + expected outputs) are committed.
Nothing is written into /root/reference (sys.dont_write_bytecode).

What is captured:
  camera.npz    utils/camera_utils.py:129-139 Camera matrix properties at Replica and
                12-Scenes intrinsics, 3 poses each (+ getProjectionMatrix2, graphics_utils.py:72-93)
  cov3d.npz     GaussianModel.build_covariance_from_scaling_rotation (gaussian_model.py:72-76,
                general_utils.py:114-148) on 64 seeded (scale, quaternion) pairs
  sh.npz        eval_sh degrees 0..3 + 0.5 + clamp (sh_utils.py:55-118, gaussian_renderer/__init__.py:85-90)
  boundary.npz  the kwargs / settings the UNMODIFIED render() passes to
                diff_gauss.GaussianRasterizer for a seeded 1000-Gaussian model
                (gaussian_renderer/__init__.py:13-141), recorded with a stub rasterizer
  loss.npz      get_loss_mapping_rgbd + BCE marker loss value and dL/dcolor, dL/ddepth on a
                seeded 48x64 image (utils/utils.py:55-82, train_gaussians.py:38-42)
"""
import json
import math
import os
import sys
import types

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))  # our diff_gauss / simple_knn
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.overrides import TorchFunctionMode  # noqa: E402


class CudaToCpu(TorchFunctionMode):
    """Redirect every device='cuda' request to the CPU; .cuda() becomes a no-op."""

    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        dev = kwargs.get("device", None)
        if dev is not None and str(dev).startswith("cuda"):
            kwargs["device"] = "cpu"
        if func is torch.Tensor.cuda:
            return args[0]
        if func is torch.Tensor.to:
            args = tuple("cpu" if isinstance(a, str) and a.startswith("cuda") else a for a in args)
        return func(*args, **kwargs)


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def main():
    # third-party modules the reference imports but this path never calls
    stub("cv2")
    o3d = stub("open3d")
    stub("plyfile", PlyData=object, PlyElement=object)
    stub("tinycudann")
    stub("models")
    stub("models.decoders", FeatureDecoder=object)
    del o3d

    with CudaToCpu():
        from gaussian_splatting.utils.graphics_utils import getProjectionMatrix2
        from gaussian_splatting.utils.sh_utils import eval_sh
        from gaussian_splatting.scene.gaussian_model import GaussianModel
        from utils.camera_utils import Camera
        import gaussian_splatting.gaussian_renderer as gr
        from utils.utils import get_loss_mapping_rgbd

        g = torch.Generator().manual_seed(1234)

        # ---------------- camera ----------------
        cams = {}
        intr = {"replica": (320.0, 320.0, 319.5, 239.5, 640, 480), "scenes12": (572.0, 572.0, 320.0, 240.0, 640, 480)}
        for name, (fx, fy, cx, cy, W, H) in intr.items():
            proj = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=fx, fy=fy, cx=cx, cy=cy, W=W, H=H).transpose(0, 1)
            fovx, fovy = 2 * math.atan(W / (2 * fx)), 2 * math.atan(H / (2 * fy))
            for k in range(3):
                q = torch.randn(4, generator=g)
                q = q / q.norm()
                r, x, y, z = q.tolist()
                R = torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)],
                                  [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
                                  [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]])
                t = torch.randn(3, generator=g)
                T = torch.eye(4)
                T[:3, :3] = R
                T[:3, 3] = t
                cam = Camera(k, None, None, T, proj, fx, fy, cx, cy, fovx, fovy, H, W, None, None, device="cpu")
                key = f"{name}_{k}"
                cams[key + "_R"] = R.numpy()
                cams[key + "_t"] = t.numpy()
                cams[key + "_view"] = cam.world_view_transform.numpy()
                cams[key + "_fullproj"] = cam.full_proj_transform.numpy()
                cams[key + "_campos"] = cam.camera_center.contiguous().numpy()
                cams[key + "_proj"] = proj.numpy()
                cams[key + "_intr"] = np.array([fx, fy, cx, cy, W, H, math.tan(fovx * 0.5), math.tan(fovy * 0.5)])
        np.savez_compressed(os.path.join(HERE, "camera.npz"), **cams)

        # ---------------- 3D covariance ----------------
        cfg = {"Training": {"primitive_reg": True}}
        gm = GaussianModel(0, config=cfg)
        scales = torch.exp(-3.0 + 0.7 * torch.randn(64, 3, generator=g))
        quats = torch.randn(64, 4, generator=g)
        quats_n = quats / quats.norm(dim=1, keepdim=True)
        cov_a = gm.build_covariance_from_scaling_rotation(scales, 1.0, quats_n)
        cov_b = gm.build_covariance_from_scaling_rotation(scales, 1.7, quats_n)
        np.savez_compressed(os.path.join(HERE, "cov3d.npz"), scales=scales.numpy(), quats=quats_n.numpy(),
                            cov_mod1=cov_a.numpy(), cov_mod1p7=cov_b.numpy())

        # ---------------- SH ----------------
        sh = {}
        dirs = torch.randn(128, 3, generator=g)
        dirs = dirs / dirs.norm(dim=1, keepdim=True)
        coeff = 0.6 * torch.randn(128, 16, 3, generator=g)
        sh["dirs"] = dirs.numpy()
        sh["coeff"] = coeff.numpy()
        for deg in range(4):
            M = (deg + 1) ** 2
            shs_view = coeff[:, :M].transpose(1, 2)  # [P,3,M] as render() builds it
            rgb = torch.clamp_min(eval_sh(deg, shs_view, dirs) + 0.5, 0.0)
            sh[f"rgb_deg{deg}"] = rgb.numpy()
        np.savez_compressed(os.path.join(HERE, "sh.npz"), **sh)

        # ---------------- boundary record: unmodified render() ----------------
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
        P = 1000
        fx, fy, cx, cy, W, H = intr["replica"]
        proj = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=fx, fy=fy, cx=cx, cy=cy, W=W, H=H).transpose(0, 1)
        T = torch.eye(4)
        T[:3, 3] = torch.tensor([0.1, -0.05, 0.3])
        fovx, fovy = 2 * math.atan(W / (2 * fx)), 2 * math.atan(H / (2 * fy))
        cam = Camera(0, None, None, T, proj, fx, fy, cx, cy, fovx, fovy, H, W, None, None, device="cpu")
        z = 0.8 + 4.0 * torch.rand(P, generator=g)
        xyz = torch.stack([(2 * torch.rand(P, generator=g) - 1) * z, (2 * torch.rand(P, generator=g) - 1) * 0.75 * z, z], 1)
        opt = types.SimpleNamespace(percent_dense=0.01, position_lr_init=0.00016, position_lr_final=0.0000016,
                                    position_lr_delay_mult=0.01, position_lr_max_steps=30000, feature_lr=0.0025,
                                    opacity_lr=0.05, marker_lr=0.05, kp_score_lr=0.05, scaling_lr=0.001,
                                    rotation_lr=0.001)
        gm.init_lr(6.0)
        gm.training_setup(opt)   # as SplatLoc.__init__ does (train_gaussians.py:66-69)
        feats = torch.zeros(P, 3, 1)
        feats[:, :, 0] = torch.randn(P, 3, generator=g)
        gm.extend_from_pcd(xyz.clone(), feats, torch.log(0.03 * torch.exp(0.4 * torch.randn(P, 3, generator=g))),
                           torch.randn(P, 4, generator=g), torch.randn(P, 1, generator=g),
                           torch.rand(P, 1, generator=g), torch.rand(P, 1, generator=g))
        pipe = types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False)
        bgc = torch.tensor([0.0, 0.0, 0.0])
        out = gr.render(cam, gm, pipe, bgc)
        kw, rs = record["kwargs"], record["settings"]
        meta = {"kwargs": {k: (None if v is None else {"shape": list(v.shape), "dtype": str(v.dtype),
                                                        "contiguous": bool(v.is_contiguous()),
                                                        "requires_grad": bool(v.requires_grad)})
                           for k, v in kw.items()},
                "settings": {k: (getattr(rs, k) if not torch.is_tensor(getattr(rs, k)) else
                                 {"shape": list(getattr(rs, k).shape), "dtype": str(getattr(rs, k).dtype),
                                  "contiguous": bool(getattr(rs, k).is_contiguous())})
                             for k in rs._fields},
                "output_keys": sorted(out.keys()),
                "output_shapes": {k: list(v.shape) for k, v in out.items() if torch.is_tensor(v)}}
        arrays = {k: v.detach().numpy() for k, v in kw.items() if v is not None}
        for k in ("bg", "viewmatrix", "projmatrix", "campos"):
            arrays["rs_" + k] = getattr(rs, k).detach().contiguous().numpy()
        np.savez_compressed(os.path.join(HERE, "boundary.npz"), meta=json.dumps(meta), **arrays)

        # ---------------- losses ----------------
        Hs, Ws = 48, 64
        image = torch.rand(3, Hs, Ws, generator=g, requires_grad=True)
        depth = (0.5 + 3 * torch.rand(1, Hs, Ws, generator=g)).requires_grad_(True)
        marker = torch.randn(Hs, Ws, generator=g, requires_grad=True)
        gt_img = torch.rand(3, Hs, Ws, generator=g)
        gt_img[:, :4] = 0.0  # below rgb_boundary_threshold
        gt_depth = (0.5 + 3 * torch.rand(Hs, Ws, generator=g)).numpy()
        gt_depth[:, :5] = 0.0
        kp = (torch.rand(Hs, Ws, generator=g) > 0.8)
        vp = types.SimpleNamespace(original_image=gt_img, depth=gt_depth)
        cfg2 = {"Training": {"rgb_boundary_threshold": 0.01}}
        loss = get_loss_mapping_rgbd(cfg2, image, depth, vp)
        bce = torch.nn.functional.binary_cross_entropy(torch.sigmoid(marker.view(-1)), kp.view(-1).float(), reduction="mean")
        (loss + bce).backward()
        np.savez_compressed(os.path.join(HERE, "loss.npz"), image=image.detach().numpy(), depth=depth.detach().numpy(),
                            marker=marker.detach().numpy(), gt_image=gt_img.numpy(), gt_depth=gt_depth,
                            kp=kp.numpy(), loss=np.array([loss.item(), bce.item()]),
                            dL_dimage=image.grad.numpy(), dL_ddepth=depth.grad.numpy(), dL_dmarker=marker.grad.numpy())
    print("golden fixtures written to", HERE)
    for f in sorted(os.listdir(HERE)):
        print(" ", f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
