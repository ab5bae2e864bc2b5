"""The mapping-loss oracle (oracle/losses.py) against the fixtures recorded from the reference's
own loss functions + autograd (tests/golden/mapping_loss.npz, loss.npz).  CPU only."""
import os

import numpy as np
import pytest

from oracle import losses

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load_case(name):
    d = np.load(os.path.join(GOLD, "mapping_loss.npz"))
    return {k[len(name) + 1:]: d[k] for k in d.files if k.startswith(name + "_")}


def check(o, c):
    assert abs(o["loss_rgbd"] - c["loss"][0]) <= 2e-6 * abs(c["loss"][0])
    assert abs(o["loss_bce"] - c["loss"][1]) <= 2e-6 * abs(c["loss"][1])
    for k in ("dL_dimage", "dL_ddepth", "dL_dmarker"):
        ref = c[k].astype(np.float64)
        assert np.abs(o[k].reshape(ref.shape) - ref).max() <= 2e-6 * np.abs(ref).max() + 1e-12, k


@pytest.mark.parametrize("name", ["exposure", "initialization"])
def test_mapping_loss_matches_reference(name):
    c = load_case(name)
    exp = name == "exposure"
    o = losses.mapping_loss(c["image"], c["depth"], c["marker"], c["gt_image"], c["gt_depth"], c["kp"], 0.01,
                            float(c["exposure"][0]) if exp else None, float(c["exposure"][1]) if exp else None)
    check(o, c)
    if exp:
        assert abs(o["dL_dexposure_a"] - c["dL_dexposure"][0]) <= 1e-5 * abs(c["dL_dexposure"][0])
        assert abs(o["dL_dexposure_b"] - c["dL_dexposure"][1]) <= 1e-5 * abs(c["dL_dexposure"][1])
    # masked-out pixels and exact zeros of the L1 argument carry no gradient
    assert np.all(o["dL_dimage"][:, :5, :] == 0) and np.all(o["dL_ddepth"][:, :, :7] == 0)
    if not exp:
        assert np.all(o["dL_dimage"][:, 10, :8] == 0) and np.all(c["dL_dimage"][:, 10, :8] == 0)


def test_first_generation_fixture():
    d = np.load(os.path.join(GOLD, "loss.npz"))
    o = losses.mapping_loss(d["image"], d["depth"], d["marker"], d["gt_image"], d["gt_depth"], d["kp"], 0.01)
    check(o, {"loss": d["loss"], "dL_dimage": d["dL_dimage"], "dL_ddepth": d["dL_ddepth"], "dL_dmarker": d["dL_dmarker"]})


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_refinement_loss_matches_reference(name):
    """(1 - lambda) L1 + lambda (1 - SSIM): value, both terms and the image gradient of the
    reference's l1_loss / ssim + autograd (loss_utils.py:21-102, train_gaussians.py:283-285)."""
    d = np.load(os.path.join(GOLD, "refinement_loss.npz"))
    o = losses.refinement_loss(d[name + "_image"], d[name + "_gt"], float(d["lambda_dssim"]))
    l1, ssim, loss = d[name + "_terms"]
    assert abs(o["l1"] - l1) <= 2e-6 * l1 and abs(o["ssim"] - ssim) <= 5e-6 * ssim and abs(o["loss"] - loss) <= 5e-6 * loss
    ref = d[name + "_dL_dimage"].astype(np.float64)
    assert np.abs(o["dL_dimage"] - ref).max() <= 2e-4 * np.abs(ref).max()
    assert np.all(o["dL_dimage"][:, -1, :3] * 0 == 0)


def test_gaussian_window_is_the_reference_window():
    w = losses.gaussian_window()
    assert w.dtype == np.float32 and abs(float(w.sum()) - 1.0) < 1e-6 and w.argmax() == 5 and len(w) == 11


def test_eval_metrics_match_reference_eval_rendering():
    """oracle.losses.eval_metrics against tests/golden/eval_rendering.npz: the per-frame PSNR / SSIM the reference's own
    `psnr` / `ssim` produced inside the restated loop body of utils/eval_utils.py:44-52 (clamp, per-element mask)."""
    d = np.load(os.path.join(GOLD, "eval_rendering.npz"))
    for k in range(3):
        o = losses.eval_metrics(d[f"view{k}_render"], d[f"view{k}_gt"])
        assert o["count"] == int(d[f"view{k}_mask_count"])
        assert abs(o["psnr"] - d["psnr"][k]) <= 2e-5 * abs(d["psnr"][k]), (k, o["psnr"], d["psnr"][k])
        # (the reference's ssim runs five float32 convolutions; sigma = E[x^2] - mu^2 cancels in float32)
        assert abs(o["ssim"] - d["ssim"][k]) <= 1e-5, (k, o["ssim"], d["ssim"][k])
        assert (d[f"view{k}_render"] > 1.0).any()                  # the clamp matters in this fixture
        assert (d[f"view{k}_gt"] == 0).any()                       # and so does the mask
    assert abs(float(d["mean_psnr"]) - d["psnr"].mean()) < 1e-9
