"""Key-frame insertion of SplatLoc on the device (SURVEY.md §8a row a10's only consumer).

`create_pcd_from_image` / `create_pcd_from_image_and_depth_score` restate
GaussianModel.create_pcd_from_image (gaussian_model.py:118-131) and create_pcd_from_image_and_depth_score (:170-217,
with creat_pcsd_from_mask :133-168) — the code that turns a key-frame's RGB-D image and SuperPoint score map into new
Gaussians and the ONLY caller of `simple_knn._C.distCUDA2` (:206) — without the reference's round trip through numpy:
the image, depth and score maps stay on the device, the 3-NN distances come from the HIP kernel behind
include/splatraster.h (`splatknn_dist2`), and `splatloc_amd.densify.extend_from_pcd` appends the rows with their
Adam-state surgery in one launch.  Same outputs as the reference (tests/test_gpu_keyframe.py against
tests/golden/keyframe.npz, recorded from the reference's own functions).

The reference down-samples the non-key pixels with `np.random.choice(n_points, n_samples)` (host RNG, with
replacement); here the draw is given as `sample_idx` (tests: the recorded draw), comes from `generator`, or — the default —
from a counter-based host generator keyed by `(seed, kf_id)`: every replica of a frame-parallel job (one process per GPU,
SURVEY.md §8e) draws the SAME indices without a broadcast, like densify's split noise (round-3 advisor finding: the
device's global RNG differs from rank to rank and the replicas would diverge silently).  No CPU fallback: tensors must be
on the ROCm device.
"""
from __future__ import annotations

import torch

from .knn import distCUDA2
from .rasterizer import _require_gpu

C0 = 0.28209479177387814   # sh_utils.py: RGB2SH(rgb) = (rgb - 0.5) / C0


def _unproject(cam, rgb, depth, scores, mask, fx, fy, cx, cy, C2W):
    """creat_pcsd_from_mask (gaussian_model.py:133-168) without the down-sampling: pixels in row-major order
    (np.argwhere), camera-space point ((col - cx) d / fx, (row - cy) d / fy, d), then C2W.  float64 like numpy's
    promotion of (int64 index - python float) * float32 depth."""
    idx = torch.nonzero(mask)                       # [N,2] (row, col), row-major — np.argwhere(mask == 1)
    d = depth[mask].to(torch.float64)
    xs = (idx[:, 1].to(torch.float64) - cx) * d / fx
    ys = (idx[:, 0].to(torch.float64) - cy) * d / fy
    pc = torch.stack([xs, ys, d], dim=-1)
    R, t = C2W[:3, :3].to(torch.float64), C2W[:3, 3].to(torch.float64)
    pw = (R @ pc.T).T + t
    return pw, rgb[mask], scores[mask]


def _np_median_f32(depth: torch.Tensor) -> torch.Tensor:
    """np.median of a float32 array: the middle value, or — for an even count — the mean of the two middle values taken in
    FLOAT32 (numpy's `mean` of a float32 pair), not the float64 midpoint torch.quantile interpolates; any size."""
    flat = depth.reshape(-1).to(torch.float32)
    n = int(flat.numel())
    srt = torch.sort(flat).values
    if n % 2:
        return srt[n // 2]
    return (srt[n // 2 - 1] + srt[n // 2]) * torch.tensor(0.5, dtype=torch.float32, device=flat.device)


def _keyed_draw(n_points: int, n_samples: int, seed: int, kf_id: int) -> torch.Tensor:
    """`n_samples` indices in [0, n_points) with replacement from a host generator keyed by (seed, kf_id): the same on
    every rank, independent of any global RNG state."""
    g = torch.Generator()
    g.manual_seed((int(seed) * 1_000_003 + int(kf_id) * 7919 + 12345) & 0x7FFFFFFFFFFFFFFF)
    return torch.randint(0, max(n_points, 1), (n_samples,), generator=g)


def create_pcd_from_image_and_depth_score(gaussians, cam, rgb, depth, scores, sample_idx=None, generator=None, seed: int = 0,
                                          kf_id: int = -1):
    """gaussian_model.py:170-217.  rgb: uint8 [H,W,3] device tensor; depth: float32 [H,W]; scores: [H,W].
    Returns (fused_point_cloud, features, scales, rots, opacities, markers, kp_scores) like the reference."""
    _require_gpu(depth, "depth")
    dev = depth.device
    cfg = gaussians.config["Dataset"]
    downsample_factor = cfg["pcd_downsample"]
    point_size = cfg["point_size"]
    if cfg.get("adaptive_pointsize", False):
        # min(0.05, point_size * np.median(depth)): np.median averages the two middle values of an even count IN FLOAT32
        med = _np_median_f32(depth).to(torch.float64)
        point_size = torch.clamp_max(point_size * med, 0.05)
    rgbf = rgb.to(torch.float64) / 255.0
    scores = scores.to(dev)
    kp_mask = (depth > 0.0) & (scores > 0.005)
    non_kp_mask = (depth > 0.0) & (scores <= 0.005)
    W2C = cam.W2C.to(device=dev, dtype=torch.float32)
    C2W = torch.linalg.inv(W2C)                      # np.linalg.inv(W2C) on the float32 pose
    fx, fy, cx, cy = float(cam.fx), float(cam.fy), float(cam.cx), float(cam.cy)
    kp_xyz, kp_rgb, kp_score = _unproject(cam, rgbf, depth, scores, kp_mask, fx, fy, cx, cy, C2W)
    nk_xyz, nk_rgb, nk_score = _unproject(cam, rgbf, depth, scores, non_kp_mask, fx, fy, cx, cy, C2W)
    if downsample_factor > 1:
        n_points = int(nk_xyz.shape[0])
        n_samples = int(n_points // downsample_factor)
        if sample_idx is None and generator is not None:
            sample_idx = torch.randint(0, max(n_points, 1), (n_samples,), device=generator.device, generator=generator)   # with replacement
        elif sample_idx is None:
            if kf_id < 0:
                # no key-frame id threaded through (the reference's signature has none, gaussian_model.py:170): key the draw by a
                # per-MODEL call counter, so that successive key-frames get independent draws like the reference's
                # np.random.choice per call — and replicas, which insert key-frames in the same order, still agree
                # (round-4 advisor finding: kf_id = -1 keyed every call alike: the same pixel ranks for every key-frame)
                n_unkeyed = int(getattr(gaussians, "_unkeyed_pcd_draws", 0))
                try:
                    gaussians._unkeyed_pcd_draws = n_unkeyed + 1
                except AttributeError:      # an object without a __dict__: no counter can be kept on it
                    raise RuntimeError("create_pcd: pass kf_id (or sample_idx / generator) for the down-sampling draw") from None
                kf_id = -1 - n_unkeyed
            sample_idx = _keyed_draw(n_points, n_samples, seed, kf_id)
        sample_idx = torch.as_tensor(sample_idx, device=dev, dtype=torch.long)
        if int(sample_idx.numel()) != n_samples:
            raise RuntimeError(f"create_pcd: sample_idx has {int(sample_idx.numel())} entries, expected {n_samples}")
        nk_xyz, nk_rgb, nk_score = nk_xyz[sample_idx], nk_rgb[sample_idx], nk_score[sample_idx]
    new_xyz = torch.cat((kp_xyz, nk_xyz), dim=0)
    new_rgb = torch.cat((kp_rgb, nk_rgb), dim=0)
    new_score = torch.cat((kp_score, nk_score), dim=0)
    fused_point_cloud = new_xyz.float()
    N = int(fused_point_cloud.shape[0])
    fused_color = (new_rgb.float() - 0.5) / C0       # RGB2SH
    features = torch.zeros((N, 3, (gaussians.max_sh_degree + 1) ** 2), dtype=torch.float32, device=dev)
    features[:, :3, 0] = fused_color
    dist2 = torch.clamp_min(distCUDA2(fused_point_cloud), 0.0000001) * point_size
    scales = torch.log(torch.sqrt(dist2.float()))[..., None]
    if not getattr(gaussians, "isotropic", False):
        scales = scales.repeat(1, 3)
    rots = torch.zeros((N, 4), device=dev)
    rots[:, 0] = 1
    x = 0.5 * torch.ones((N, 1), dtype=torch.float, device=dev)
    opacities = torch.log(x / (1 - x))               # inverse_sigmoid(0.5), general_utils.py:20-21
    markers = new_score[:, None].float()
    kp_scores = 0.5 * torch.ones((N, 1), dtype=torch.float, device=dev)
    return fused_point_cloud, features, scales, rots, opacities, markers, kp_scores


def create_pcd_from_image(gaussians, cam_info, depthmap, sample_idx=None, generator=None, seed: int = 0, kf_id: int = -1):
    """gaussian_model.py:118-131: exposure affine, clamp, uint8 colours; `depthmap` float32 [H,W] (device tensor)."""
    cam = cam_info
    image_ab = torch.exp(cam.exposure_a) * cam.original_image + cam.exposure_b
    image_ab = torch.clamp(image_ab, 0.0, 1.0)
    rgb = (image_ab * 255).byte().permute(1, 2, 0).contiguous()
    depth = torch.as_tensor(depthmap, device=rgb.device, dtype=torch.float32)
    scores = cam.kp_score
    return create_pcd_from_image_and_depth_score(gaussians, cam, rgb.detach(), depth, scores, sample_idx=sample_idx,
                                                 generator=generator, seed=seed, kf_id=kf_id)


def extend_from_pcd_seq(gaussians, cam_info, kf_id=-1, init=False, scale=2.0, depthmap=None, sample_idx=None,
                        generator=None, seed: int = 0) -> int:
    """GaussianModel.extend_from_pcd_seq (gaussian_model.py:243-248; train_gaussians.py:177) on the device.  The
    down-sampling draw: `sample_idx`, else `generator`, else keyed by `(seed, kf_id)` (identical on every replica)."""
    from .densify import extend_from_pcd
    with torch.no_grad():
        tensors = create_pcd_from_image(gaussians, cam_info, depthmap, sample_idx=sample_idx, generator=generator, seed=seed,
                                        kf_id=kf_id)
    return extend_from_pcd(gaussians, *tensors)
