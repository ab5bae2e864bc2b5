"""Generates tests/golden/refinement_loss.npz by calling the reference's own l1_loss and ssim
(gaussian_splatting/utils/loss_utils.py:21-22, 42-102) in THIS container, combined exactly as the
colour-refinement loop does (train_gaussians.py:283-285, lambda_dssim = 0.2):
    loss = (1 - lambda) * l1_loss(image, gt) + lambda * (1 - ssim(image, gt))
and recording the value, the two terms and the autograd gradient w.r.t. the image, on seeded
frames of two sizes (one not a multiple of the 16x16 tile).  Only the fixture is committed.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    mg.stub("cv2")
    with mg.CudaToCpu():
        from gaussian_splatting.utils.loss_utils import l1_loss, ssim
        g = torch.Generator().manual_seed(99)
        out = {}
        lam = 0.2
        for name, (C, H, W) in (("a", (3, 48, 64)), ("b", (3, 37, 53)), ("c", (1, 9, 7))):
            gt = torch.rand(C, H, W, generator=g)
            # a render that resembles the target (as in training) plus structure-free regions
            image = (gt + 0.15 * torch.randn(C, H, W, generator=g)).clamp(0, 1)
            image[:, : H // 4] = 0.5
            with torch.no_grad():
                image[:, -1, :3] = gt[:, -1, :3]            # exact zeros of the L1 argument
            image.requires_grad_(True)
            l1 = l1_loss(image, gt)
            s = ssim(image, gt)
            loss = (1.0 - lam) * l1 + lam * (1.0 - s)
            loss.backward()
            out[name + "_image"] = image.detach().numpy().copy()
            out[name + "_gt"] = gt.numpy().copy()
            out[name + "_terms"] = np.array([l1.item(), s.item(), loss.item()])
            out[name + "_dL_dimage"] = image.grad.numpy().copy()
        out["lambda_dssim"] = np.array(lam)
    path = os.path.join(HERE, "refinement_loss.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), {k: out[k] for k in out if k.endswith("terms")})


if __name__ == "__main__":
    main()
