"""CPU oracle of SplatLoc's per-view mapping loss and its gradients — numpy
restatement of the reference's in-tree Python.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing on the product path).
Parity status: PINNED by tests/golden/mapping_loss.npz and tests/golden/loss.npz (values and
autograd gradients recorded from the reference's own functions;
tests/golden/make_golden_losses.py, make_golden.py).

  loss = mean |m_rgb x - m_rgb gt| + mean |m_d depth - m_d gt_depth| + mean BCE(sigmoid(marker), kp)
     x      = exp(exposure_a) image + exposure_b        (image itself at initialization)   utils/utils.py:55-61
     m_rgb  = sum_c gt[c] > rgb_boundary_threshold,  m_d = gt_depth > 0.01               utils/utils.py:74-75
     L1 terms: means over 3 H W and H W elements                                          utils/utils.py:77-81
     BCE: torch.nn.functional.binary_cross_entropy (log clamped at -100), mean            train_gaussians.py:38-42
  summed per view exactly as SplatLoc.map does (train_gaussians.py:217-218).
"""
from __future__ import annotations

import numpy as np


def mapping_loss(image, depth, marker, gt_image, gt_depth, kp, rgb_boundary_threshold, exposure_a=None,
                 exposure_b=None):
    """Returns dict(loss_rgbd, loss_bce, dL_dimage, dL_ddepth, dL_dmarker, dL_dexposure_a, dL_dexposure_b);
    the forward sums in float32 like torch, the gradients are exact (float64)."""
    f32 = np.float32
    image, depth, marker = np.asarray(image, f32), np.asarray(depth, f32).reshape(1, *np.shape(marker)), np.asarray(marker, f32)
    gt_image, gt_depth = np.asarray(gt_image, f32), np.asarray(gt_depth, f32).reshape(depth.shape)
    y = np.asarray(kp).astype(f32)
    H, W = marker.shape
    use_exp = exposure_a is not None
    ea = f32(np.exp(f32(exposure_a))) if use_exp else f32(1.0)
    eb = f32(exposure_b) if use_exp else f32(0.0)
    x = ea * image + eb if use_exp else image
    m_rgb = (gt_image.sum(axis=0) > f32(rgb_boundary_threshold)).reshape(1, H, W).astype(f32)
    m_d = (gt_depth > f32(0.01)).astype(f32)
    diff = x * m_rgb - gt_image * m_rgb
    diffd = depth * m_d - gt_depth * m_d
    l_rgbd = np.abs(diff).mean(dtype=np.float64) + np.abs(diffd).mean(dtype=np.float64)
    with np.errstate(over="ignore"):
        p = (1.0 / (1.0 + np.exp(-marker))).astype(f32)
    with np.errstate(divide="ignore"):
        logp = np.maximum(np.log(p), f32(-100.0))
        log1p = np.maximum(np.log(f32(1.0) - p), f32(-100.0))
    l_bce = (-(y * logp + (1.0 - y) * log1p)).mean(dtype=np.float64)
    n_rgb, n = 3.0 * H * W, 1.0 * H * W
    g_x = np.sign(diff).astype(np.float64) * m_rgb / n_rgb
    p64 = p.astype(np.float64)
    g_marker = (p64 - y) / np.maximum((1.0 - p64) * p64, 1e-12) * (p64 * (1.0 - p64)) / n
    return dict(loss_rgbd=l_rgbd, loss_bce=l_bce, dL_dimage=g_x * float(ea),
                dL_ddepth=np.sign(diffd).astype(np.float64) * m_d / n, dL_dmarker=g_marker,
                dL_dexposure_a=float((g_x * image).sum() * float(ea)) if use_exp else 0.0,
                dL_dexposure_b=float(g_x.sum()) if use_exp else 0.0)


# ---------------------------------------------------------------------------------------------
# colour-refinement loss: (1 - lambda) L1 + lambda (1 - SSIM)
#   l1_loss   gaussian_splatting/utils/loss_utils.py:21-22
#   ssim      gaussian_splatting/utils/loss_utils.py:42-102 (11x11 Gaussian window, sigma 1.5, zero
#             padding, C1 = 0.01^2, C2 = 0.03^2, mean over all elements)
#   combined  train_gaussians.py:283-285 (lambda_dssim)
# Parity status: PINNED by tests/golden/refinement_loss.npz (tests/golden/make_golden_ssim.py).
# ---------------------------------------------------------------------------------------------
def gaussian_window(size: int = 11, sigma: float = 1.5) -> np.ndarray:
    """loss_utils.py:42-49: float32 1-D window, normalised."""
    g = np.array([np.exp(-((x - size // 2) ** 2) / float(2 * sigma ** 2)) for x in range(size)], dtype=np.float32)
    return g / g.sum(dtype=np.float32)


def _blur(a: np.ndarray, w: np.ndarray) -> np.ndarray:
    """Zero-padded separable correlation of every [H,W] plane with w (x) w (float64)."""
    r = len(w) // 2
    H, W = a.shape[-2:]
    pad = np.pad(a.astype(np.float64), [(0, 0)] * (a.ndim - 2) + [(r, r), (r, r)])
    tmp = sum(w[k] * pad[..., :, k:k + W] for k in range(len(w)))
    return sum(w[k] * tmp[..., k:k + H, :] for k in range(len(w)))


def refinement_loss(image, gt, lambda_dssim: float = 0.2):
    """Returns dict(l1, ssim, loss, dL_dimage) — float64 restatement (the window itself is the
    reference's float32 window)."""
    x, y = np.asarray(image, np.float64), np.asarray(gt, np.float64)
    w = gaussian_window().astype(np.float64)
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    mu1, mu2 = _blur(x, w), _blur(y, w)
    s1, s2, s12 = _blur(x * x, w), _blur(y * y, w), _blur(x * y, w)
    sig1, sig2, sig12 = s1 - mu1 * mu1, s2 - mu2 * mu2, s12 - mu1 * mu2
    A, B = 2 * mu1 * mu2 + C1, 2 * sig12 + C2
    Cc, D = mu1 * mu1 + mu2 * mu2 + C1, sig1 + sig2 + C2
    m = A * B / (Cc * D)
    n = x.size
    l1 = np.abs(x - y).mean()
    ssim = m.mean()
    # d m / d(mu1, s1, s12) with s1 = blur(x^2), s12 = blur(x y) held as independent inputs
    dm_dmu1 = (2 * mu2 * (B - A) * Cc * D - A * B * 2 * mu1 * (D - Cc)) / (Cc * D) ** 2
    dm_ds1 = -A * B / (Cc * D * D)
    dm_ds12 = 2 * A / (Cc * D)
    dssim_dx = (_blur(dm_dmu1, w) + 2 * x * _blur(dm_ds1, w) + y * _blur(dm_ds12, w)) / n
    grad = (1.0 - lambda_dssim) * np.sign(x - y) / n - lambda_dssim * dssim_dx
    return dict(l1=l1, ssim=ssim, loss=(1.0 - lambda_dssim) * l1 + lambda_dssim * (1.0 - ssim), dL_dimage=grad)


def eval_metrics(render, gt):
    """Per-frame metrics of the reference's eval_rendering (utils/eval_utils.py:45-52):

        image = torch.clamp(rendering, 0.0, 1.0);  mask = gt_image > 0                       (per ELEMENT)
        psnr  = 20 log10(1 / sqrt(mean((image[mask] - gt[mask])^2)))      gaussian_splatting/utils/image_utils.py:19-21
        ssim  = ssim(image, gt)                                           loss_utils.py:61-102 (11x11 window, zero padding)

    float64 restatement (the window is the reference's float32 window); pinned by tests/golden/eval_rendering.npz, recorded
    from the reference's own render / psnr / ssim.  Returns dict(psnr, ssim, mse, count)."""
    x = np.clip(np.asarray(render, np.float64), 0.0, 1.0)
    y = np.asarray(gt, np.float64)
    mask = y > 0
    count = int(mask.sum())
    mse = float(((x[mask] - y[mask]) ** 2).mean()) if count else float("nan")
    w = gaussian_window().astype(np.float64)
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    mu1, mu2 = _blur(x, w), _blur(y, w)
    sig1, sig2, sig12 = _blur(x * x, w) - mu1 * mu1, _blur(y * y, w) - mu2 * mu2, _blur(x * y, w) - mu1 * mu2
    m = ((2 * mu1 * mu2 + C1) * (2 * sig12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (sig1 + sig2 + C2))
    return dict(psnr=20.0 * np.log10(1.0 / np.sqrt(mse)) if count else float("nan"), ssim=float(m.mean()), mse=mse, count=count)
