"""Generates tests/golden/mapping_loss.npz by calling the reference's own loss functions in THIS
container: get_loss_mapping with the exposure affine (utils/utils.py:55-82) +
get_loss_marker (train_gaussians.py:38-42), exactly as SplatLoc.map sums them per view
(train_gaussians.py:217-218), on a seeded 60x80 frame; records the loss value and the autograd
gradients w.r.t. image, depth, marker, exposure_a, exposure_b.  The `exposure_` case uses a float
key-point score map (soft BCE targets, as on real data), the `initialization_` case a bool mask.
Only the fixture is committed.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    for m in ("cv2", "open3d", "tinycudann", "models"):
        mg.stub(m)
    with mg.CudaToCpu():
        from utils.utils import get_loss_mapping
        g = torch.Generator().manual_seed(77)
        H, W = 60, 80
        out = {}
        for name, init in (("exposure", False), ("initialization", True)):
            image = torch.rand(3, H, W, generator=g, requires_grad=True)
            depth = (0.5 + 3 * torch.rand(1, H, W, generator=g)).requires_grad_(True)
            marker = (3 * torch.randn(H, W, generator=g)).requires_grad_(True)
            with torch.no_grad():
                marker[0, :4] = torch.tensor([60.0, -60.0, 120.0, -120.0])   # saturating logits (BCE log clamp)
            gt_img = torch.rand(3, H, W, generator=g)
            gt_img[:, :5] = 0.0                                  # below rgb_boundary_threshold
            gt_depth = (0.5 + 3 * torch.rand(H, W, generator=g)).numpy()
            gt_depth[:, :7] = 0.0                                # invalid depth
            with torch.no_grad():
                image[:, 10, :8] = gt_img[:, 10, :8]             # exact zeros of the L1 argument (sign(0) = 0)
            if init:
                kp = torch.rand(H, W, generator=g) > 0.8             # a bool mask: hard 0 / 1 targets
            else:
                # what the dataset really hands over (utils/dataset.py:94 `*_score.npy`, camera_utils.py:75):
                # a continuous score map in [0, 1] -> SOFT BCE targets (train_gaussians.py:40 `.float()`)
                kp = torch.rand(H, W, generator=g) ** 4
                kp[1, :6] = torch.tensor([0.0, 1.0, 0.005, 0.5, 1e-6, 0.999])
            a = torch.tensor([0.13], requires_grad=True)
            b = torch.tensor([-0.04], requires_grad=True)
            vp = types.SimpleNamespace(original_image=gt_img, depth=gt_depth, exposure_a=a, exposure_b=b)
            cfg = {"Training": {"rgb_boundary_threshold": 0.01}}
            # train_gaussians.py:38-42 (get_loss_marker), restated here because importing
            # train_gaussians pulls OpenGL / open3d GUI modules
            pred = torch.sigmoid(marker.view(-1))
            bce = torch.nn.functional.binary_cross_entropy(pred, kp.view(-1).float(), reduction="mean")
            lm = get_loss_mapping(cfg, image, depth, vp, None, initialization=init)
            (lm + bce).backward()
            pre = name + "_"
            out.update({pre + "image": image.detach().numpy().copy(), pre + "depth": depth.detach().numpy().copy(),
                        pre + "marker": marker.detach().numpy().copy(), pre + "gt_image": gt_img.numpy().copy(),
                        pre + "gt_depth": gt_depth.copy(), pre + "kp": kp.numpy().copy(),
                        pre + "exposure": np.array([a.item(), b.item()], np.float32),
                        pre + "loss": np.array([lm.item(), bce.item()]),
                        pre + "dL_dimage": image.grad.numpy().copy(), pre + "dL_ddepth": depth.grad.numpy().copy(),
                        pre + "dL_dmarker": marker.grad.numpy().copy(),
                        pre + "dL_dexposure": np.array([0.0 if a.grad is None else a.grad.item(),
                                                        0.0 if b.grad is None else b.grad.item()])})
    path = os.path.join(HERE, "mapping_loss.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))
    print({k: (v.shape, float(np.abs(v).max())) for k, v in out.items() if "dL" in k or "loss" in k})


if __name__ == "__main__":
    main()
