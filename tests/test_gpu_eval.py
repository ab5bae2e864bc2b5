"""`eval_rendering` on the device (splatloc_amd.evaluation; utils/eval_utils.py:22-72 of the reference — BASELINE config 4's
stand-in) against tests/golden/eval_rendering.npz: per-frame PSNR / SSIM recorded from the reference's own `render` /
`psnr` / `ssim` inside the restated loop body (clamp to [0, 1], PSNR over the ELEMENTS where gt > 0, SSIM over the frame),
with the CPU oracle standing in for the un-vendored rasterizer.  Needs an MI355X."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model_and_frames(d, dev):
    from splatloc_amd.camera import PinholeCamera
    par = lambda a: torch.nn.Parameter(torch.from_numpy(a).to(dev).contiguous().requires_grad_(True))  # noqa: E731
    gm = types.SimpleNamespace(active_sh_degree=0, max_sh_degree=0)
    for k, a in (("xyz", "_xyz"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("opacity", "_opacity"),
                 ("kp_score", "_kp_score"), ("scaling", "_scaling"), ("rotation", "_rotation")):
        setattr(gm, a, par(d["model_" + k]))
    fx, fy, cx, cy, W, H = (float(v) for v in d["intr"][:6])
    frames, gts = [], []
    for k in range(3):
        T = torch.from_numpy(d[f"view{k}_T"])
        frames.append(PinholeCamera(int(W), int(H), fx, fy, cx, cy, T[:3, :3], T[:3, 3]).to(dev))
        gts.append(torch.from_numpy(d[f"view{k}_gt"]).to(dev))
    return gm, frames, gts


def test_eval_metrics_kernel_on_the_recorded_renders(golden_dir):
    """The metrics kernel alone on the RECORDED (un-clamped) renders: PSNR / SSIM / mask count of the reference."""
    from splatloc_amd.evaluation import eval_metrics
    d = np.load(os.path.join(golden_dir, "eval_rendering.npz"))
    for k in range(3):
        out = eval_metrics(torch.from_numpy(d[f"view{k}_render"]).to(DEV), torch.from_numpy(d[f"view{k}_gt"]).to(DEV)).cpu()
        assert int(out[3]) == int(d[f"view{k}_mask_count"])
        assert abs(float(out[0]) - d["psnr"][k]) <= 2e-5 * d["psnr"][k], (k, float(out[0]), d["psnr"][k])
        assert abs(float(out[1]) - d["ssim"][k]) <= 1e-5, (k, float(out[1]), d["ssim"][k])


@pytest.mark.parametrize("window", [1, 2, 5])
def test_eval_rendering_loop_matches_reference_recording(golden_dir, window):
    """The whole loop: forward-only windows through the HIP rasterizer + the metrics kernel; `window` = 1 is the
    reference's frame-by-frame loop, 2 leaves a ragged last window, 5 holds all frames in one launch sequence."""
    from splatloc_amd.evaluation import eval_rendering
    d = np.load(os.path.join(golden_dir, "eval_rendering.npz"))
    dev = torch.device(DEV)
    gm, frames, gts = _model_and_frames(d, dev)
    pipe = types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False)
    bg = torch.zeros(3, device=dev)
    out = eval_rendering(frames, gm, gts, pipe, bg, window=window)
    assert out["frames"] == 3 and out["mean_lpips"] is None
    # the image is the rasterizer's (<= 1e-4 of the oracle's): PSNR ~ 25 dB moves by < 1e-3 dB, SSIM by < 1e-4
    assert np.abs(np.array(out["psnr"]) - d["psnr"]).max() <= 2e-3, (out["psnr"], d["psnr"])
    assert np.abs(np.array(out["ssim"]) - d["ssim"]).max() <= 1e-4, (out["ssim"], d["ssim"])
    assert abs(out["mean_psnr"] - float(d["mean_psnr"])) <= 2e-3 and abs(out["mean_ssim"] - float(d["mean_ssim"])) <= 1e-4
    # an invalid frame (gt None: the reference's `valid == False`) is skipped
    out2 = eval_rendering(frames, gm, [gts[0], None, gts[2]], pipe, bg, window=window)
    assert out2["frames"] == 2 and abs(out2["psnr"][1] - d["psnr"][2]) <= 2e-3
    assert all(p.grad is None for p in (gm._xyz, gm._opacity))     # forward only: nothing was differentiated


def test_eval_metrics_edge_cases():
    from splatloc_amd.evaluation import eval_metrics
    g = torch.Generator().manual_seed(5)
    gt = torch.rand(3, 33, 47, generator=g).to(DEV)        # not a multiple of the 16 x 16 tile
    same = eval_metrics(gt.clone(), gt).cpu()
    assert torch.isinf(same[0]) and same[0] > 0 and abs(float(same[1]) - 1.0) < 1e-6 and float(same[2]) == 0.0
    empty = eval_metrics(gt, torch.zeros_like(gt)).cpu()    # empty mask: the reference's mean over nothing is NaN
    assert torch.isnan(empty[0]) and float(empty[3]) == 0.0
    with pytest.raises(RuntimeError):
        eval_metrics(gt.cpu(), gt.cpu())
