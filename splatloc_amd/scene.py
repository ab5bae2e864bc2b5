"""The whole reconstruction schedule of SplatLoc as ONE run (train_gaussians.py:310-355 `do_recon`):

    for every key-frame:   add_next_kf -> extend_from_pcd_seq (key-frame insertion, distCUDA2)       :173-177, 332
                           map(iters = mapping_itr_num): 10 optimisation steps on 5 random views,   :179-267
                           with densify_and_prune / reset_opacity_nonvisible on their schedules
    color_refinement():    26 000 single-view iterations                                             :269-297
    save_gaussians(...):   point_cloud.ply                                                           :355

chained on the device-side pieces of this package (keyframe.extend_from_pcd_seq, training.map_step,
training.color_refinement_step, ply.save_ply) with P growing from zero: allocator growth, window buffers re-sized after
every densification, Adam-state surgery after surgery.  Every piece is pinned on its own by a reference-recorded
fixture; this module is what runs them for thousands of iterations in sequence
(tests/test_gpu_scene.py, `bench.py --stage scene`).

`SceneModel` holds exactly the attributes of the reference's GaussianModel (gaussian_model.py:35-70, 250-300) that those
pieces read; the reference's own GaussianModel object works in its place.  `synthetic_keyframes` renders RGB-D key-frames
of a ground-truth Gaussian set with the rasterizer itself (there is no dataset in the container): a moving camera, metric
depth = depth / alpha where alpha is high, a sparse key-point score map.  No CPU fallback anywhere.
"""
from __future__ import annotations

import math
import random
import time
import types

import torch

from .camera import PinholeCamera
from .densify import ATTR, GROUPS
from .keyframe import extend_from_pcd_seq
from .training import LAST_STEP_INFO, color_refinement_step, map_step

# configs/replica_nerf/base_config.yaml (the values train_gaussians.py reads)
DEFAULT_CONFIG = {
    "Dataset": {"pcd_downsample": 64, "pcd_downsample_init": 32, "adaptive_pointsize": True, "point_size": 0.05},
    "Training": {"mapping_itr_num": 10, "gaussian_update_every": 150, "gaussian_update_offset": 50, "gaussian_th": 0.7,
                 "gaussian_extent": 1.0, "gaussian_reset": 2001, "size_threshold": 20, "window_size": 5,
                 "rgb_boundary_threshold": 0.01, "primitive_reg": True},
    "opt_params": {"position_lr_init": 0.00016, "position_lr_final": 0.0000016, "position_lr_delay_mult": 0.01,
                   "position_lr_max_steps": 30000, "feature_lr": 0.0025, "opacity_lr": 0.05, "marker_lr": 0.05,
                   "kp_score_lr": 0.05, "scaling_lr": 0.001, "rotation_lr": 0.001, "percent_dense": 0.01, "lambda_dssim": 0.2,
                   "densify_grad_threshold": 0.0002},
}


class SceneModel:
    """The attribute layout of GaussianModel (gaussian_model.py:35-70) with `training_setup` (:250-300) on an EMPTY model:
    8 parameter groups, Adam(lr = 0, eps = 1e-15) — `splatloc_amd.optim.Adam` (one launch per step) or torch.optim.Adam."""

    def __init__(self, config=None, device="cuda", sh_degree: int = 0, fused_adam: bool = True, spatial_lr_scale: float = 6.0):
        from torch import nn
        cfg = config or DEFAULT_CONFIG
        self.config = cfg
        self.active_sh_degree, self.max_sh_degree = 0, sh_degree
        self.primitive_reg = bool(cfg["Training"].get("primitive_reg", True))
        self.isotropic = False
        dev = torch.device(device)
        opt = cfg["opt_params"]
        self.percent_dense = opt["percent_dense"]
        self.spatial_lr_scale = spatial_lr_scale          # SplatLoc: gaussians.init_lr(6.0), train_gaussians.py:68
        K = (sh_degree + 1) ** 2
        shapes = {"xyz": (0, 3), "f_dc": (0, 1, 3), "f_rest": (0, K - 1, 3), "opacity": (0, 1), "marker": (0, 1), "kp_score": (0, 1),
                  "scaling": (0, 3), "rotation": (0, 4)}
        for k in GROUPS:
            setattr(self, ATTR[k], nn.Parameter(torch.zeros(shapes[k], device=dev).requires_grad_(True)))
        lr = {"xyz": opt["position_lr_init"] * spatial_lr_scale, "f_dc": opt["feature_lr"], "f_rest": opt["feature_lr"] / 20.0,
              "opacity": opt["opacity_lr"], "marker": opt["marker_lr"], "kp_score": opt["kp_score_lr"],
              "scaling": opt["scaling_lr"] * spatial_lr_scale, "rotation": opt["rotation_lr"]}
        groups = [{"params": [getattr(self, ATTR[k])], "lr": lr[k], "name": k} for k in GROUPS]
        if fused_adam:
            from .optim import Adam
            self.optimizer = Adam(groups, lr=0.0, eps=1e-15)
        else:
            self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        self.lr_init = opt["position_lr_init"] * spatial_lr_scale
        self.lr_final = opt["position_lr_final"] * spatial_lr_scale
        self.lr_delay_mult = opt["position_lr_delay_mult"]
        self.max_steps = opt["position_lr_max_steps"]
        self.xyz_gradient_accum = torch.zeros((0, 1), device=dev)
        self.denom = torch.zeros((0, 1), device=dev)
        self.max_radii2D = torch.zeros((0,), device=dev)

    @property
    def num_points(self) -> int:
        return int(self._xyz.shape[0])


class KeyFrame(PinholeCamera):
    """What the pieces read of utils/camera_utils.py's Camera: the matrices, `original_image [3,H,W]`, `depth [H,W]`,
    `kp_score [H,W]`, `exposure_a / exposure_b`, `W2C`, intrinsics, `uid`."""

    def __init__(self, uid, W, H, fx, fy, cx, cy, R, t, device):
        super().__init__(W, H, fx, fy, cx, cy, R, t)
        self.to(device)
        self.uid = uid
        T = torch.eye(4)
        T[:3, :3], T[:3, 3] = R, t
        self.W2C = T.to(device)
        self.exposure_a = torch.zeros(1, device=device, requires_grad=True)
        self.exposure_b = torch.zeros(1, device=device, requires_grad=True)
        self.original_image = self.depth = self.kp_score = None


def synthetic_keyframes(n_frames: int, W: int = 640, H: int = 480, P_truth: int = 60_000, seed: int = 0, device="cuda",
                        fx: float = None):
    """RGB-D key-frames of a synthetic room: `P_truth` ground-truth Gaussians on the walls / floor of a box and on a few
    blobs inside it, seen by a camera that pans and translates through `n_frames` poses (Replica intrinsics by default:
    fx = fy = W / 2, principal point (W - 1) / 2, (H - 1) / 2).  Rendered by the rasterizer itself, forward only.
    Returns (keyframes, truth) with truth = dict of the ground-truth tensors."""
    from .rasterizer import GaussianRasterizationSettings, rasterize_window
    dev = torch.device(device)
    g = torch.Generator().manual_seed(1000 + seed)
    fx = fx or W / 2.0
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    # a 6 x 3 x 8 m box around the origin: points on its five visible faces + three blobs
    n_wall = int(P_truth * 0.85)
    face = torch.randint(0, 5, (n_wall,), generator=g)
    u, v = torch.rand(n_wall, generator=g), torch.rand(n_wall, generator=g)
    X, Y, Z = 3.0, 1.5, 6.0
    xyz = torch.zeros(n_wall, 3)
    xyz[face == 0] = torch.stack([(2 * u - 1) * X, (2 * v - 1) * Y, torch.full_like(u, Z)], 1)[face == 0]          # back wall
    xyz[face == 1] = torch.stack([torch.full_like(u, -X), (2 * v - 1) * Y, u * Z], 1)[face == 1]                    # left
    xyz[face == 2] = torch.stack([torch.full_like(u, X), (2 * v - 1) * Y, u * Z], 1)[face == 2]                     # right
    xyz[face == 3] = torch.stack([(2 * u - 1) * X, torch.full_like(u, Y), v * Z], 1)[face == 3]                     # floor (y down)
    xyz[face == 4] = torch.stack([(2 * u - 1) * X, torch.full_like(u, -Y), v * Z], 1)[face == 4]                    # ceiling
    n_blob = P_truth - n_wall
    centres = torch.tensor([[-1.2, 0.6, 3.0], [0.9, 0.2, 4.2], [0.1, 0.9, 2.4]])
    blob = centres[torch.randint(0, 3, (n_blob,), generator=g)] + 0.25 * torch.randn(n_blob, 3, generator=g)
    xyz = torch.cat([xyz, blob])
    # a smooth colour field + per-face tint so that neighbouring key-frames agree on what they see
    rgb = 0.5 + 0.25 * torch.stack([torch.sin(1.3 * xyz[:, 0] + 0.4 * xyz[:, 2]), torch.sin(1.7 * xyz[:, 1] + 0.9),
                                     torch.cos(0.8 * xyz[:, 2] - 0.5 * xyz[:, 0])], 1)
    rgb = (rgb + 0.08 * torch.randn(P_truth, 3, generator=g)).clamp(0.02, 0.98)
    spacing = math.sqrt((2 * X * 2 * Y + 2 * Z * 2 * Y * 2 + 2 * X * Z * 2) / n_wall)
    scales = (0.9 * spacing * torch.exp(0.15 * torch.randn(P_truth, 3, generator=g))).clamp(0.01, 0.2)
    rots = torch.nn.functional.normalize(torch.randn(P_truth, 4, generator=g), dim=1)
    opac = torch.full((P_truth, 1), 0.97)
    truth = {"xyz": xyz.to(dev), "rgb": rgb.to(dev), "scales": scales.to(dev), "rots": rots.to(dev), "opac": opac.to(dev)}
    bg = torch.zeros(3, device=dev)
    frames = []
    for k in range(n_frames):
        a = (k / max(n_frames - 1, 1) - 0.5) * 0.9                 # pan of +-26 degrees
        R = torch.tensor([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]], dtype=torch.float32)
        c = torch.tensor([1.2 * math.sin(2.2 * a), 0.1 * math.cos(3 * a), 0.3 + 0.8 * k / max(n_frames - 1, 1)])     # camera centre
        frames.append(KeyFrame(k, W, H, fx, fx, cx, cy, R, -(R @ c), dev))
    with torch.no_grad():
        for a in range(0, n_frames, 5):
            chunk = frames[a:a + 5]
            settings = [GaussianRasterizationSettings(H, W, f.tanfovx, f.tanfovy, bg, 1.0, f.world_view_transform,
                                                      f.full_proj_transform, 0, f.camera_center, False, False) for f in chunk]
            carriers = [torch.zeros_like(truth["xyz"]) for _ in chunk]
            outs = rasterize_window(settings, truth["xyz"], carriers, truth["rgb"], truth["opac"], scales=truth["scales"],
                                    rotations=truth["rots"])
            for f, (color, depth, alpha, _) in zip(chunk, outs):
                solid = alpha[0] > 0.6
                f.original_image = torch.where(solid[None], color.clamp(0, 1), torch.zeros_like(color)).contiguous()
                f.depth = torch.where(solid, depth[0] / alpha[0].clamp_min(1e-3), torch.zeros_like(depth[0])).contiguous()
                # a sparse key-point score map (SuperPoint's role, utils/dataset.py:94): high at ~0.5 % of the pixels
                gk = torch.Generator().manual_seed(7000 + 31 * seed + f.uid)
                score = torch.rand(H, W, generator=gk) ** 6 * 0.004
                hot = torch.rand(H, W, generator=gk) < 0.005
                score[hot] = 0.2 + 0.7 * torch.rand(int(hot.sum()), generator=gk)
                f.kp_score = score.to(dev)
    return frames, truth


def load_depth(config, viewpoint):
    """SplatLoc.load_depth (train_gaussians.py:298-308): the observed depth with invalid-RGB pixels zeroed."""
    thr = config["Training"]["rgb_boundary_threshold"]
    valid = viewpoint.original_image.sum(dim=0) > thr
    return torch.where(valid, viewpoint.depth, torch.zeros_like(viewpoint.depth))


def do_recon(gaussians, keyframes, pipe=None, background=None, config=None, refine_iterations: int = 26000, seed: int = 0,
             batched: bool = True, group=None, on_event=None, distributed: bool = True) -> dict:
    """SplatLoc.do_recon (train_gaussians.py:310-355) on `keyframes` (the reference: every `kf_interval`-th dataset frame).
    `batched = False` renders every window — and every refinement iteration — as the reference does: one `render()` per view
    through the drop-in autograd.Function (`render_path="per-view"` of training.map_step / color_refinement_step) instead of
    one graph-free launch sequence per window; `stats["render_paths"]` records which paths the map steps took.  The random draws the reference takes from global RNGs (`torch.randperm` of the window,
    `random.randint` of the refinement view, `np.random.choice` of the key-frame down-sampling) come from generators seeded
    by `seed`, identical on every rank of a frame-parallel job.  `distributed = False`: no collective even when a process
    group exists — every rank reconstructs a scene of its own (one scene per GPU: /root/reference/replica.sh).  Returns counters
    and timings; the model is updated in place."""
    cfg = config or gaussians.config
    tr, opt = cfg["Training"], cfg["opt_params"]
    dev = gaussians._xyz.device
    pipe = pipe or types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False)
    background = background if background is not None else torch.zeros(3, device=dev)     # train_gaussians.py:70
    cameras_extent = 6.0                                                                   # train_gaussians.py:72
    dens = dict(grad_threshold=opt["densify_grad_threshold"], min_opacity=tr["gaussian_th"],
                extent=cameras_extent * tr["gaussian_extent"], size_threshold=tr["size_threshold"],
                every=tr["gaussian_update_every"], offset=tr["gaussian_update_offset"])
    rng_w = torch.Generator().manual_seed(50_000 + seed)
    rng_r = random.Random(60_000 + seed)
    viewpoints = {}
    stats = {"rows_after_keyframe": [], "densify_rows": [], "map_iterations": 0, "refine_iterations": 0, "resets": 0}
    iteration_count = 0
    t0 = time.perf_counter()
    path = "auto" if batched else "per-view"
    stats["render_paths"] = set()
    for kf_id, viewpoint in enumerate(keyframes):
        viewpoints[kf_id] = viewpoint
        extend_from_pcd_seq(gaussians, viewpoint, kf_id=kf_id, depthmap=load_depth(cfg, viewpoint), seed=seed)
        stats["rows_after_keyframe"].append(gaussians.num_points if hasattr(gaussians, "num_points") else int(gaussians._xyz.shape[0]))
        stack = list(viewpoints.values())
        for _ in range(tr["mapping_itr_num"]):
            iteration_count += 1
            idx = torch.randperm(len(stack), generator=rng_w)[:tr["window_size"]]          # train_gaussians.py:195
            rows = int(gaussians._xyz.shape[0])
            map_step([stack[i] for i in idx], gaussians, pipe, background, cfg, iteration_count, densify=dens,
                     gaussian_reset=tr["gaussian_reset"], seed=seed, group=group, render_path=path, distributed=distributed)
            stats["render_paths"].add(LAST_STEP_INFO.get("render_path"))
            if int(gaussians._xyz.shape[0]) != rows:
                stats["densify_rows"].append([iteration_count, rows, int(gaussians._xyz.shape[0])])
            if tr["gaussian_reset"] and iteration_count % tr["gaussian_reset"] == 0:
                stats["resets"] += 1
        if on_event:
            on_event("keyframe", kf_id, gaussians)
    stats["map_iterations"] = iteration_count
    torch.cuda.synchronize(dev)
    stats["map_seconds"] = time.perf_counter() - t0
    t1 = time.perf_counter()
    keys = list(viewpoints.keys())
    for iteration in range(1, refine_iterations + 1):                                       # train_gaussians.py:269-297
        cam = viewpoints[keys[rng_r.randint(0, len(keys) - 1)]]
        color_refinement_step(cam, gaussians, pipe, background, opt["lambda_dssim"], iteration,
                              primitive_reg=bool(tr.get("primitive_reg", True)), render_path=path)
        if on_event and iteration % 500 == 0:
            on_event("refine", iteration, gaussians)
    if refine_iterations and distributed:
        # one view per step does not shard: every rank refined its own replica redundantly, and float-atomic
        # rounding lets redundant replicas drift in the last bits — rank 0's state becomes everybody's again (one broadcast)
        from .frame_parallel import broadcast_model
        stats["refine_broadcast_bytes"] = broadcast_model(gaussians, src=0, group=group)
    torch.cuda.synchronize(dev)
    stats["refine_iterations"] = refine_iterations
    stats["refine_seconds"] = time.perf_counter() - t1
    stats["render_paths"] = sorted(p for p in stats["render_paths"] if p)
    stats["rows_final"] = int(gaussians._xyz.shape[0])
    stats["peak_memory_bytes"] = int(torch.cuda.max_memory_allocated(dev))
    return stats


def state_digest(gaussians) -> str:
    """sha256 over every parameter, Adam moment, step counter and statistic — replica consistency checks."""
    import hashlib
    h = hashlib.sha256()
    for grp in gaussians.optimizer.param_groups:
        p = grp["params"][0]
        h.update(p.detach().cpu().contiguous().numpy().tobytes())
        st = gaussians.optimizer.state.get(p, None)
        if st:
            h.update(st["exp_avg"].cpu().contiguous().numpy().tobytes())
            h.update(st["exp_avg_sq"].cpu().contiguous().numpy().tobytes())
            h.update(str(float(st["step"])).encode())
        h.update(repr(grp["lr"]).encode())
    for k in ("xyz_gradient_accum", "denom", "max_radii2D"):
        h.update(getattr(gaussians, k).cpu().contiguous().numpy().tobytes())
    return h.hexdigest()
