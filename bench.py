#!/usr/bin/env python3
"""bench.py — fwd+bwd frames/s of the MI355X-native Gaussian rasterizer (BASELINE.json metric).

One "step" = one optimisation step of SplatLoc.map as far as the rasterizer is concerned
(train_gaussians.py:187-229): `--views` = 5 frames (the reference's window_size, configs/*/
base_config.yaml: window_size: 5) seen from 5 DIFFERENT cameras (the reference draws 5 different
key-frames, train_gaussians.py:195), each one forward + one backward pass of the rasterizer hot
path over the synthetic workload S2 (500k Gaussians, 1920x1080, 35 channels: RGB + 32 feature
channels, + depth + alpha), inputs resident in HBM, gradients accumulated over the window, the
per-view densification statistics updated (train_gaussians.py:238-245).  By default the window goes
through the window-batched C-ABI sequence (splatloc_amd.rasterize_window: ONE preprocess / depth sort /
tile sort / compositing grid for the 5 views, per-view results bit-identical to 5 calls);
`--no-window` runs the reference's loop of 5 diff_gauss.GaussianRasterizer calls instead.

N > 1 ranks (one process per GPU, RCCL over xGMI), full scene replica per rank:
  --scaling weak   (default) every rank renders its OWN window of 5 views (different per rank);
                   value = 5 N frames per step / max-over-ranks step time;
  --scaling strong the ONE window of 5 views is dealt round-robin to the ranks
                   (frame_parallel.shard_views) — exactly the reference step, faster.
In both the accumulated parameter gradients are SUM-all-reduced ONCE per step inside the timed
region and the densification statistics of the step are synchronised (SUM / MAX) so that every
replica would densify identically.

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline`
(dominant kernel, HIP events measured live on the launch stream) and `cpu_baseline`
(the CPU oracle timed on this host's cores on the same workload: warm-up + 3 frames, median/min,
plus a single-core figure from a bounded row band).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BREAKDOWN_STEPS = 20   # untimed steps that collect the per-stage table (and bring the clocks up)


def algorithmic_bytes(P, R, W, H, C, tiles):
    """Compulsory bytes of the LINEAGE formulation: per stage and per fwd+bwd frame."""
    tile_bits = max(1, (tiles - 1).bit_length())
    n_pass = (32 + tile_bits + 7) // 8
    stage = {
        "preprocess": P * (44 + 4 * C + 60),
        "depth_sort": 0,  # not part of the lineage formulation (its sort is the R-sized one)
        "scan": P * 8,
        "emit": R * 12,
        "tile_sort": R * 24 * n_pass,
        "ranges": tiles * 8,  # clearing the table; the R * 8 boundary scan rides in the payload kernel
        "composite_fwd": R * (32 + 4 * C) + W * H * (4 * C + 16),
        "composite_bwd": R * (32 + 4 * C) + W * H * (4 * C + 16) + P * (28 + 4 * C),
        "preprocess_bwd": P * (100 + 4 * C + 56 + 4 * C),
        "payload": 0,  # not in the lineage
    }
    frame = P * (296 + 16 * C) + R * (20 + 24 * n_pass + 2 * (32 + 4 * C)) + W * H * (8 * C + 32)
    return stage, frame, n_pass


def actual_bytes(P, V, R, W, H, C, tiles):
    """Compulsory bytes of THIS implementation's stages (every array once per pass that needs it; L2 /
    Infinity-Cache hits, atomics and re-reads not counted) — the figure a stage's GB/s is quoted on.
    DESIGN.md §5 / HISTORY.md §4.2 derive each line."""
    CP = (C + 3) & ~3
    mo = C if ((C & 15) + 7 <= 16) else ((C + 15) & ~15)
    grow = (mo + 7 + 15) & ~15                     # gacc_row_floats(C)
    tile_bits = max(1, (tiles - 1).bit_length())
    tile_passes = (tile_bits + 7) // 8
    return {
        "preprocess": P * (44 + 32 + 4 + 4 + 4 + 4),           # means/opacity/scale/quat in; record, tiles, radii, key, id out
        "depth_sort": P * (4 + 4 * 16),                        # histogram read + 4 passes x (key, id) read + write
        "scan": P * 12,                                        # tiles_touched (gathered), permutation, offsets
        "emit": P * 20 + R * 8,                                # offsets, rect sources in; (tile id, gaussian id) out
        "tile_sort": R * 20 * tile_passes,                     # per pass: histogram 4 + scatter 8 in + 8 out
        "ranges": tiles * 8,
        "payload": R * (8 + 32 + 36) + ((P * 4 * (C + CP)) if C % 4 else 0),   # ids + gathered records in; records + packed word out; padded feature table
        "composite_fwd": R * 36 + V * 4 * CP + W * H * (4 * C + 16),           # packed word + record per entry; staged rows; planes out
        # round 4: the back-to-front backward reads dL/dout (4 C + 8) + n_contrib, final_T (8) per pixel, NOT the forward's planes;
        # records only for its quadrant's candidates (upper bound: every entry)
        "composite_bwd": R * 36 + V * 4 * CP + W * H * (4 * C + 16) + 2 * V * 4 * grow,
        "preprocess_bwd": P * (4 * grow + 44 + 32 + 7) + P * (4 * C + 12 + 12 + 4 + 12 + 16),
    }


def cpu_baseline(workload, frames=3):
    """the oracle (OpenMP build, every host core) on fwd+bwd frames of the same workload —
    1 warm-up frame, then `frames` timed ones, median and min — plus a single-core figure from a bounded
    sample: the non-OpenMP build on the full P-sized front end and a 64-row band of the image,
    extrapolated to the full height."""
    from oracle import oracle
    from splatloc_amd.synthetic import WORKLOADS, make_workload
    from tests.helpers import oracle_backward, oracle_forward
    sc = make_workload(workload)
    H = WORKLOADS[workload]["H"]
    oracle.build()
    oracle.set_row_band()
    times = []
    for k in range(frames + 1):
        t0 = time.perf_counter()
        f = oracle_forward(sc, omp=True)
        t1 = time.perf_counter()
        oracle_backward(f, sc, omp=True)
        t2 = time.perf_counter()
        if k:
            times.append((t2 - t0, t1 - t0, t2 - t1))
    times.sort()
    med, best = times[len(times) // 2], times[0]
    # single core: a band of `rows` image rows in the middle of the frame (the front end is run in full)
    rows = 64
    y0 = (H - rows) // 2 // 16 * 16
    try:
        oracle.set_row_band(y0, y0 + rows)
        t0 = time.perf_counter()
        f1 = oracle_forward(sc, omp=False)
        t1 = time.perf_counter()
        oracle_backward(f1, sc, omp=False)
        t2 = time.perf_counter()
        oracle.set_row_band(y0, y0)          # empty band: the P-sized front end + preprocess backward alone
        t3 = time.perf_counter()
        f0 = oracle_forward(sc, omp=False)
        oracle_backward(f0, sc, omp=False)
        t4 = time.perf_counter()
    finally:
        oracle.set_row_band()
    front = t4 - t3
    band = max((t2 - t0) - front, 1e-9)
    single = front + band * (H / rows)
    return {"value": 1.0 / med[0], "unit": "frames/s", "cores": oracle.num_threads(True), "kind": "port",
            "sample": f"{frames} fwd+bwd frames of {workload} after 1 warm-up frame, oracle/splat_oracle.c built with "
                      f"-fopenmp: median {med[0]:.2f} s (fwd {med[1]:.2f} + bwd {med[2]:.2f}), min {best[0]:.2f} s",
            "min_frames_per_s": 1.0 / best[0],
            "single_core": {"value": 1.0 / single, "unit": "frames/s", "cores": 1,
                            "sample": f"non-OpenMP build: full front end + preprocess backward ({front:.2f} s) + compositing "
                                      f"fwd+bwd of image rows [{y0}, {y0 + rows}) ({band:.2f} s), extrapolated x {H}/{rows}"},
            "host_cpu_count": os.cpu_count()}


def cpu_baseline_torch_dense(workload="S0", rows=8, threads=None):
    """BASELINE config 1's "PyTorch-CPU autograd reference" (BASELINE.md §2 item 2), timed: tests/torch_dense_ref.render_dense
    — float32 tensor algebra + torch.autograd on the host cores, every pixel of a band against every visible Gaussian —
    fwd + bwd of `rows` image rows in the middle of the frame (the full S0 frame is 3e9 pairs: ~12 GB per intermediate),
    extrapolated to the frame's height.  A bounded sample of the same workload; reported beside the oracle's figure."""
    from splatloc_amd.synthetic import WORKLOADS, make_workload
    from tests.torch_dense_ref import render_dense
    wl = WORKLOADS[workload]
    sc = make_workload(workload)
    cam = sc.camera
    H, W = wl["H"], wl["W"]
    if threads:
        torch.set_num_threads(threads)
    y0 = (H - rows) // 2
    leaf = lambda t: t.detach().clone().float().requires_grad_(True)  # noqa: E731
    times = []
    for k in range(2):      # one warm-up band, one timed
        m3, col, op, sca, rot = (leaf(t) for t in (sc.means3D, sc.features, sc.opacities, sc.scales, sc.rotations))
        t0 = time.perf_counter()
        color, depth, alpha, _ = render_dense(H, W, cam.tanfovx, cam.tanfovy, sc.bg, m3, op, cam.world_view_transform,
                                              cam.full_proj_transform, colors_precomp=col, scales=sca, rotations=rot,
                                              rows=(y0, y0 + rows))
        loss = ((color * sc.dL_dcolor[:, y0:y0 + rows]).sum() + (depth * sc.dL_ddepth[:, y0:y0 + rows]).sum()
                + (alpha * sc.dL_dalpha[:, y0:y0 + rows]).sum())
        t1 = time.perf_counter()
        loss.backward()
        t2 = time.perf_counter()
        times.append((t2 - t0, t1 - t0, t2 - t1))
    band = times[-1]
    frame_s = band[0] * H / rows
    return {"value": 1.0 / frame_s, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"tests/torch_dense_ref.render_dense in float32 + torch.autograd on the host: fwd+bwd of image rows "
                      f"[{y0}, {y0 + rows}) of {workload} ({band[0]:.2f} s: fwd {band[1]:.2f} + bwd {band[2]:.2f}; second of two "
                      f"runs), extrapolated x {H}/{rows}"}


def bench_activations(args, dev):
    """--stage activations: the fused parameter activations + SH / feature packing (SURVEY.md
    §8f-1, splatloc_amd.fused.activate_pack) forward + backward on the S2 parameter shapes
    (500k Gaussians, SH degree 0, 32 extra feature columns), next to the same arithmetic as the
    reference's chain of torch ops on the same GPU.  HBM-bound: achieved GB/s vs the 8 TB/s peak."""
    from splatloc_amd.fused import activate_pack
    P, E = 500_000, 32
    g = torch.Generator().manual_seed(7)
    leaf = lambda *s: torch.randn(*s, generator=g).to(dev).requires_grad_(True)  # noqa: E731
    xyz, f_dc, scaling, rotation, opacity, extra = leaf(P, 3), leaf(P, 1, 3), leaf(P, 3), leaf(P, 4), leaf(P, 1), leaf(P, E)
    f_rest = torch.zeros(P, 0, 3, device=dev, requires_grad=True)
    campos = torch.zeros(3, device=dev)
    G = [torch.randn(P, 3, generator=g).to(dev), torch.randn(P, 4, generator=g).to(dev),
         torch.randn(P, 1, generator=g).to(dev), torch.randn(P, 3 + E, generator=g).to(dev)]
    leaves = [xyz, f_dc, scaling, rotation, opacity, extra]

    def fused():
        outs = activate_pack(xyz, f_dc, f_rest, scaling, rotation, opacity, extra=extra, campos=campos)
        torch.autograd.backward(outs, G)

    def composed():  # gaussian_model.py:78-105 + gaussian_renderer/__init__.py:84-102 at SH degree 0
        feats = torch.cat((f_dc, f_rest), dim=1)
        shs_view = feats.transpose(1, 2).view(-1, 3, 1)
        rgb = torch.clamp_min(0.28209479177387814 * shs_view[..., 0] + 0.5, 0.0)
        outs = (torch.exp(scaling), torch.nn.functional.normalize(rotation), torch.sigmoid(opacity),
                torch.cat((rgb, extra), dim=1))
        torch.autograd.backward(outs, G)

    def time_it(fn):
        for _ in range(args.warmup):
            for t in leaves:
                t.grad = None
            fn()
        torch.cuda.synchronize(dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record()
        for _ in range(args.steps):
            for t in leaves:
                t.grad = None
            fn()
        b.record()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / args.steps * 1e3, a.elapsed_time(b) / args.steps

    wall_f, gpu_f = time_it(fused)
    wall_c, gpu_c = time_it(composed)

    # the two kernels alone, launched back to back through the C ABI (no autograd, no allocation):
    # this is the figure held against the HBM roofline
    import ctypes as C
    from splatloc_amd import _native
    lib = _native.load()
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    with torch.no_grad():
        o = [torch.empty(P, 3, device=dev), torch.empty(P, 4, device=dev), torch.empty(P, 1, device=dev),
             torch.empty(P, 3 + E, device=dev)]
        d = [torch.empty_like(t) for t in (f_dc, scaling, rotation, opacity, extra)]
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def kernels():
        _native.check(lib.splatraster_activate_forward(P, 1, 0, 3, E, p(xyz), p(f_dc), None, p(scaling), p(rotation),
                                                       p(opacity), p(extra), p(campos), p(o[0]), p(o[1]), p(o[2]),
                                                       p(o[3]), st), "activate_forward")
        _native.check(lib.splatraster_activate_backward(P, 1, 0, 3, E, p(xyz), p(f_dc), None, p(scaling), p(rotation),
                                                        p(opacity), p(campos), p(G[0]), p(G[1]), p(G[2]), p(G[3]),
                                                        None, p(d[0]), None, p(d[1]), p(d[2]), p(d[3]), p(d[4]), st),
                      "activate_backward")

    _, gpu_k = time_it(kernels)
    fwd_bytes = P * ((12 + 12 + 16 + 4 + 4 * E) + (12 + 16 + 4 + 4 * (3 + E)))
    bwd_bytes = P * ((12 + 12 + 16 + 4) + (12 + 16 + 4 + 4 * (3 + E)) + (12 + 12 + 16 + 4 + 4 * E))
    ach = (fwd_bytes + bwd_bytes) / (gpu_k * 1e-3) / 1e9
    print(json.dumps({
        "metric": "activation + SH/feature packing fwd+bwd passes/s (SURVEY 8f-1 stage; NOT the BASELINE metric)",
        "value": round(1e3 / wall_f, 1), "unit": "passes/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall_f, 4), "higher_is_better": True, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"P={P} Gaussians, SH degree 0, {E} extra feature columns (S2 parameter shapes)"},
        "roofline": {"bound": "hbm", "kernel": "activate_fwd_kernel + activate_bwd_kernel", "achieved": round(ach, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                     "kernels_ms": round(gpu_k, 4), "algorithmic_bytes": fwd_bytes + bwd_bytes},
        "through_autograd": {"ms_per_step": round(wall_f, 4), "note": "host-bound: Python autograd + allocation"},
        "torch_ops_same_gpu": {"ms_per_step": round(wall_c, 4), "speedup_wall": round(wall_c / wall_f, 2)},
    }), flush=True)


def bench_loss(args, dev):
    """--stage loss: the fused per-view mapping loss + gradient (SURVEY.md §8f-2,
    splatloc_amd.losses) at 1920x1080 next to the reference's chain of torch ops on the same GPU."""
    import ctypes as C
    from splatloc_amd import _native
    from splatloc_amd.losses import mapping_loss_tensors
    H, W = 1080, 1920
    g = torch.Generator().manual_seed(3)
    image = torch.rand(3, H, W, generator=g).to(dev).requires_grad_(True)
    depth = (0.5 + 3 * torch.rand(1, H, W, generator=g)).to(dev).requires_grad_(True)
    marker = (2 * torch.randn(H, W, generator=g)).to(dev).requires_grad_(True)
    gt_image, gt_depth = torch.rand(3, H, W, generator=g).to(dev), (0.5 + 3 * torch.rand(H, W, generator=g)).to(dev)
    kp = (torch.rand(H, W, generator=g) ** 4).to(dev)   # float score map: soft BCE targets (train_gaussians.py:40)
    a = torch.tensor([0.05], device=dev, requires_grad=True)
    b = torch.tensor([0.02], device=dev, requires_grad=True)
    leaves = [image, depth, marker, a, b]

    def fused():
        mapping_loss_tensors(image, depth, marker, gt_image, gt_depth, kp, 0.01, a, b).backward()

    def composed():  # utils/utils.py:55-82 + train_gaussians.py:38-42
        x = torch.exp(a) * image + b
        m = (gt_image.sum(dim=0) > 0.01).view(*depth.shape)
        md = (gt_depth[None] > 0.01).view(*depth.shape)
        loss = torch.abs(x * m - gt_image * m).mean() + torch.abs(depth * md - gt_depth[None] * md).mean() \
            + torch.nn.functional.binary_cross_entropy(torch.sigmoid(marker.view(-1)), kp.view(-1).float(),
                                                       reduction="mean")
        loss.backward()

    lib = _native.load()
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    with torch.no_grad():
        gi, gd, gm = torch.empty_like(image), torch.empty_like(depth), torch.empty_like(marker)
        out = torch.empty(4, device=dev)
        ex = torch.cat((a, b)).detach()
        k8 = kp.to(torch.float32).contiguous()
        ws = torch.empty(lib.splatraster_mapping_loss_workspace_bytes(H * W), dtype=torch.uint8, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def kernels():
        _native.check(lib.splatraster_mapping_loss(H * W, p(image), p(depth), p(marker), p(gt_image), p(gt_depth), p(k8),
                                                   C.c_float(0.01), p(ex), p(gi), p(gd), p(gm), p(out), p(ws), st),
                      "mapping_loss")

    def time_it(fn):
        for _ in range(args.warmup):
            for t in leaves:
                t.grad = None
            fn()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(args.steps):
            for t in leaves:
                t.grad = None
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / args.steps * 1e3, e0.elapsed_time(e1) / args.steps

    wall_f, _ = time_it(fused)
    wall_c, _ = time_it(composed)
    _, gpu_k = time_it(kernels)

    # colour-refinement loss: (1 - 0.2) L1 + 0.2 (1 - SSIM), train_gaussians.py:283-285
    from splatloc_amd.losses import refinement_loss
    win = torch.tensor([math.exp(-((x - 5) ** 2) / 4.5) for x in range(11)])
    win = (win / win.sum()).to(dev)
    win2d = (win[:, None] @ win[None, :]).expand(3, 1, 11, 11).contiguous()

    def ssim_torch(a_, b_):   # loss_utils.py:72-102
        conv = lambda t: torch.nn.functional.conv2d(t, win2d, padding=5, groups=3)  # noqa: E731
        mu1, mu2 = conv(a_), conv(b_)
        s1, s2, s12 = conv(a_ * a_) - mu1 * mu1, conv(b_ * b_) - mu2 * mu2, conv(a_ * b_) - mu1 * mu2
        return (((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))).mean()

    def refine_fused():
        refinement_loss(image, gt_image, 0.2).backward()

    def refine_composed():
        (0.8 * torch.abs(image - gt_image).mean() + 0.2 * (1.0 - ssim_torch(image[None], gt_image[None]))).backward()

    wall_rf, gpu_rf = time_it(refine_fused)
    wall_rc, gpu_rc = time_it(refine_composed)
    nbytes = H * W * (9 * 4 + 4 + 5 * 4)
    ach = nbytes / (gpu_k * 1e-3) / 1e9
    print(json.dumps({
        "metric": "mapping loss + gradient passes/s at 1920x1080 (SURVEY 8f-2 stage; NOT the BASELINE metric)",
        "value": round(1e3 / wall_f, 1), "unit": "passes/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall_f, 4), "higher_is_better": True, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "1920x1080 frame: L1 RGB (exposure affine, masks) + L1 depth + BCE marker"},
        "roofline": {"bound": "hbm", "kernel": "mapping_loss_kernel (+ finish)", "achieved": round(ach, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                     "kernels_ms": round(gpu_k, 4), "algorithmic_bytes": nbytes},
        "through_autograd": {"ms_per_step": round(wall_f, 4)},
        "torch_ops_same_gpu": {"ms_per_step": round(wall_c, 4), "speedup_wall": round(wall_c / wall_f, 2)},
        "refinement_loss_L1_SSIM": {"fused_ms": round(wall_rf, 4), "fused_gpu_ms": round(gpu_rf, 4),
                                    "torch_conv_chain_ms": round(wall_rc, 4), "torch_gpu_ms": round(gpu_rc, 4),
                                    "speedup_wall": round(wall_rc / wall_rf, 2)},
    }), flush=True)


def bench_map_step(args, dev):
    """--stage map_step: one optimisation step of SplatLoc.map (train_gaussians.py:187-267) on the
    S2 shapes — 5 views x (render -> per-view mapping loss), the isotropic regulariser, ONE backward,
    densification statistics, the key-primitive gradient gate + Adam over the 8 parameter groups, and
    densify_and_prune every `--densify-every` steps — with the fused device-side pieces
    (splatloc_amd.fused.render, .losses.mapping_loss / isotropic_loss, .densify, .optim.Adam) and, for
    comparison, with the reference's chains of torch ops around the same rasterizer (no densification in
    that leg: the reference's densify is ~90 boolean-mask launches with host syncs; its cost is reported
    separately for the fused path).  Secondary figure, not the BASELINE metric."""
    import types
    from splatloc_amd import GaussianRasterizationSettings, GaussianRasterizer
    from splatloc_amd.camera import PinholeCamera
    from splatloc_amd.densify import add_densification_stats_window, densify_and_prune
    from splatloc_amd.fused import render as fused_render, render_window
    from splatloc_amd.losses import isotropic_loss, mapping_loss, mapping_loss_window
    from splatloc_amd.optim import Adam as FusedAdam
    from splatloc_amd.synthetic import WORKLOADS, make_workload
    wl = WORKLOADS[args.workload]
    sc = make_workload(args.workload)
    P0, W, H, C = wl["P"], wl["W"], wl["H"], wl["C"]
    E = max(C - 3, 1)
    inv_sig = lambda p: torch.log(p / (1 - p))  # noqa: E731
    NAMES = ("xyz", "f_dc", "f_rest", "opacity", "marker", "kp_score", "scaling", "rotation")
    ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
            "marker": "_marker", "kp_score": "_kp_score", "scaling": "_scaling", "rotation": "_rotation"}
    LR = {"xyz": 1.6e-4 * 6.0, "f_dc": 2.5e-3, "f_rest": 2.5e-3 / 20, "opacity": 5e-2, "marker": 5e-2, "kp_score": 5e-2,
          "scaling": 1e-3 * 6.0, "rotation": 1e-3}

    def make_model(adam_cls, **adam_kw):
        g = torch.Generator().manual_seed(11)
        par = lambda t: torch.nn.Parameter(t.to(dev).contiguous().requires_grad_(True))  # noqa: E731
        pc = types.SimpleNamespace(
            _xyz=par(sc.means3D.clone()),
            _features_dc=par(((sc.features[:, :3] - 0.5) / 0.28209479177387814)[:, None, :].contiguous()),
            _features_rest=par(torch.zeros(P0, 0, 3)), _opacity=par(inv_sig(sc.opacities.clamp(1e-4, 1 - 1e-4))),
            _marker=par((torch.rand(P0, 1, generator=g) < 0.05).float() * torch.rand(P0, 1, generator=g) * 0.9),
            _kp_score=par(torch.rand(P0, E, generator=g)), _scaling=par(torch.log(sc.scales)),
            _rotation=par(sc.rotations.clone()), active_sh_degree=0, max_sh_degree=0, percent_dense=0.01,
            primitive_reg=True, lr_init=1.6e-4 * 6.0, lr_final=1.6e-6 * 6.0, lr_delay_mult=0.01, max_steps=30000)
        pc.optimizer = adam_cls([{"params": [getattr(pc, ATTR[k])], "lr": LR[k], "name": k} for k in NAMES], lr=0.0,
                                eps=1e-15, **adam_kw)
        pc.xyz_gradient_accum = torch.zeros(P0, 1, device=dev)
        pc.denom = torch.zeros(P0, 1, device=dev)
        pc.max_radii2D = torch.zeros(P0, device=dev)
        return pc

    g = torch.Generator().manual_seed(12)
    views = []
    for k in range(5):
        ang = torch.tensor(0.02 * (k - 2))
        R = torch.tensor([[torch.cos(ang), 0, torch.sin(ang)], [0, 1, 0], [-torch.sin(ang), 0, torch.cos(ang)]])
        cam = PinholeCamera(W, H, W / 2.0, W / 2.0, (W - 1) / 2.0, (H - 1) / 2.0, R, torch.tensor([0.01 * k, 0.0, 0.0])).to(dev)
        cam.original_image = torch.rand(3, H, W, generator=g).to(dev)
        cam.depth = (0.5 + 3 * torch.rand(H, W, generator=g)).to(dev)
        cam.kp_score = (torch.rand(H, W, generator=g) ** 4).to(dev)   # float score map (utils/dataset.py:94)
        cam.exposure_a = torch.zeros(1, device=dev, requires_grad=True)
        cam.exposure_b = torch.zeros(1, device=dev, requires_grad=True)
        views.append(cam)
    bg = torch.zeros(3, device=dev)
    pipe = types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False)
    cfg = {"Training": {"rgb_boundary_threshold": 0.01}}
    cfg_full = {"Training": {"rgb_boundary_threshold": 0.01, "primitive_reg": True}}
    from splatloc_amd.training import map_step

    def composed_render(pc, cam):   # gaussian_renderer/__init__.py:59-126 with torch ops
        rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg, 1.0,
                                           cam.world_view_transform, cam.full_proj_transform, 0, cam.camera_center,
                                           False, False)
        m2 = torch.zeros_like(pc._xyz, requires_grad=True) + 0
        m2.retain_grad()
        feats = torch.cat((pc._features_dc, pc._features_rest), dim=1)
        rgb = torch.clamp_min(0.28209479177387814 * feats.transpose(1, 2).view(-1, 3, 1)[..., 0] + 0.5, 0.0)
        img, depth, alpha, radii = GaussianRasterizer(raster_settings=rs)(
            means3D=pc._xyz, means2D=m2, shs=None, colors_precomp=torch.cat((rgb, pc._kp_score), dim=1),
            opacities=torch.sigmoid(pc._opacity), scales=torch.exp(pc._scaling),
            rotations=torch.nn.functional.normalize(pc._rotation), cov3D_precomp=None)
        return {"render": img[:3], "kp_prob": img[-1], "viewspace_points": m2, "visibility_filter": radii > 0,
                "radii": radii, "depth": depth, "opacity": alpha}

    def composed_loss(cam, image, depth, marker):   # utils/utils.py:55-82 + train_gaussians.py:38-42
        x = torch.exp(cam.exposure_a) * image + cam.exposure_b
        m = (cam.original_image.sum(dim=0) > 0.01).view(*depth.shape)
        md = (cam.depth[None] > 0.01).view(*depth.shape)
        return torch.abs(x * m - cam.original_image * m).mean() + torch.abs(depth * md - cam.depth[None] * md).mean() \
            + torch.nn.functional.binary_cross_entropy(torch.sigmoid(marker.view(-1)), cam.kp_score.view(-1).float(),
                                                       reduction="mean")

    def step(pc, fused, it, densify_ms):
        loss, pkgs = 0, []
        bw_tensors, bw_grads = [], []
        if fused and max(args.streams, 1) == 1 and not args.no_window:
            # THE PRODUCT FUNCTION: splatloc_amd.training.map_step (window launch sequence, per-view losses carrying their own
            # gradients, regulariser, one window backward — without an autograd graph since round 4 —, key gate, statistics,
            # densify_and_prune on its schedule, fused Adam, lr schedule)
            n0 = pc._xyz.shape[0]
            dens = None
            if args.densify_every and it >= 0:
                dens = dict(grad_threshold=0.0002, min_opacity=0.005, extent=6.0, size_threshold=20, every=args.densify_every,
                            offset=args.densify_every // 3)      # thresholds of configs/replica_nerf/base_config.yaml (init_gaussian_th)
            map_step(views, pc, pipe, bg, cfg_full, max(it, 0), densify=dens, seed=7)
            if pc._xyz.shape[0] != n0:
                densify_ms.append((None, None, n0, pc._xyz.shape[0]))
            for cam in views:
                cam.exposure_a.grad = cam.exposure_b.grad = None
            return
        if fused and max(args.streams, 1) == 1:
            # training.map_step's path: one launch sequence for the window, the per-view losses carry their own gradients
            # (losses.mapping_loss_window): ONE backward on the rasterizer's outputs
            pkgs, _ = render_window(views, pc, pipe, bg)
            bw_tensors, bw_grads, loss = mapping_loss_window(cfg, pkgs, views)
        elif fused:   # one activate_pack per window; the views on --streams HIP streams (HISTORY.md §11)
            pkgs, losses = render_window(views, pc, pipe, bg, streams=max(args.streams, 1),
                                         per_view=lambda k, cam, pkg: mapping_loss(cfg, pkg["render"], pkg["depth"], pkg["kp_prob"], cam))
            loss = sum(losses)
        else:
            for cam in views:
                pkg = fused_render(cam, pc, pipe, bg) if fused else composed_render(pc, cam)
                if fused:
                    loss = loss + mapping_loss(cfg, pkg["render"], pkg["depth"], pkg["kp_prob"], cam)
                else:
                    loss = loss + composed_loss(cam, pkg["render"], pkg["depth"], pkg["kp_prob"])
                pkgs.append(pkg)
        if fused and bw_tensors:
            reg = 0.01 * isotropic_loss(torch.exp(pc._scaling), pc._marker)
            bw_tensors, bw_grads = bw_tensors + [reg], bw_grads + [None]
        elif fused:      # train_gaussians.py:221-228, no .cpu() sync
            loss = loss + 0.01 * isotropic_loss(torch.exp(pc._scaling), pc._marker)
        else:
            scaling, score = torch.exp(pc._scaling), pc._marker.detach()
            mask = score.cpu().squeeze() > 0.005
            loss = loss + 0.01 * torch.abs(scaling.mean(dim=1).view(-1, 1)[mask] / (0.02 * (1 - score[mask])) - 1).mean()
        if bw_tensors:
            torch.autograd.backward(bw_tensors, bw_grads)
        else:
            loss.backward()
        with torch.no_grad():   # train_gaussians.py:231-267, gaussian_model.py:677-679
            if fused:
                pc.optimizer.set_key_gate(pc._marker, 0.005)
            else:
                key_mask = pc._marker.detach().cpu().squeeze() > 0.005
                pc._xyz.grad[key_mask] = 0
            if fused:   # the 5 views in one launch
                add_densification_stats_window([pkg["viewspace_points"].grad for pkg in pkgs], [pkg["radii"] for pkg in pkgs],
                                               pc.xyz_gradient_accum, pc.denom, pc.max_radii2D)
            for pkg in ([] if fused else pkgs):
                vis = pkg["visibility_filter"]
                pc.max_radii2D[vis] = torch.max(pc.max_radii2D[vis], pkg["radii"][vis].float())
                pc.xyz_gradient_accum[vis] += torch.norm(pkg["viewspace_points"].grad[vis, :2], dim=-1, keepdim=True)
                pc.denom[vis] += 1
            if fused and args.densify_every and it % args.densify_every == args.densify_every // 3:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                n_before = pc._xyz.shape[0]
                densify_and_prune(pc, 0.0002, 0.005, 6.0, 20, seed=7)   # thresholds of configs/replica_nerf/base_config.yaml (init_gaussian_th)
                e1.record()
                densify_ms.append((e0, e1, n_before, pc._xyz.shape[0]))
            pc.optimizer.step()
            pc.optimizer.zero_grad(set_to_none=True)
            for cam in views:
                cam.exposure_a.grad = cam.exposure_b.grad = None

    def time_it(fused):
        pc = make_model(FusedAdam) if fused else make_model(torch.optim.Adam, fused=True)
        dens = []
        for it in range(args.warmup):
            step(pc, fused, -1, dens)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for it in range(args.steps):
            step(pc, fused, it, dens)
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        return ms, [((a.elapsed_time(b) if a is not None else None), n0, n1) for a, b, n0, n1 in dens]

    ms_f, dens = time_it(True)
    saved_every = args.densify_every
    args.densify_every = 0
    ms_nd, _ = time_it(True)          # the same step on a fresh model of constant size (no densification)
    args.densify_every = saved_every
    ms_c, _ = time_it(False)
    dens_total = sum(d[0] or 0.0 for d in dens)
    print(json.dumps({
        "metric": "SplatLoc.map optimisation steps/s (5 views/step; secondary figure, NOT the BASELINE metric)",
        "value": round(1e3 / ms_f, 2), "unit": "steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_f, 3), "higher_is_better": True, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: P={P0}, {W}x{H}, C={3 + E} ([rgb | {E} kp/feature columns]) + depth + alpha; "
                               "5 views x (render + mapping loss), isotropic regulariser, one backward, densification "
                               f"stats, key gate + fused Adam over 8 groups, densify_and_prune every {args.densify_every} steps; "
                               f"{max(args.streams, 1)} HIP stream(s) per window"},
        "views_per_s": round(5e3 / ms_f, 1),
        "densify": {"calls_in_timed_region": len(dens), "rows_before_after": [[d[1], d[2]] for d in dens],
                    "note": "the model GROWS during the timed region (rows_before_after); ms_per_step_constant_size is the same step "
                            "on a fresh model without densification"},
        "ms_per_step_constant_size": round(ms_nd, 3),
        "torch_front_end_loss_adam_same_rasterizer_no_densify": {"ms_per_step": round(ms_c, 3), "speedup": round(ms_c / ms_nd, 3)},
    }), flush=True)


def bench_refine_step(args, dev):
    """--stage refine_step: iterations of SplatLoc.color_refinement (train_gaussians.py:269-297; 26 000 of the ~35 000
    rasterizer calls of a scene): ONE view per iteration — render, 0.8 L1 + 0.2 (1 - SSIM) on RGB, backward, key-primitive
    gate, max_radii2D update, Adam over the 8 groups, lr schedule — with splatloc_amd.training.color_refinement_step
    (window-of-one launch sequence, RGB-only backward, fused loss / Adam) and, for comparison, with the reference's chain
    of torch ops around the same per-view rasterizer call.  Default workload: the reference's own layout (S2-ref-layout:
    500k Gaussians, 640x480, C = 4).  Secondary figure, not the BASELINE metric."""
    import types
    from splatloc_amd import GaussianRasterizationSettings, GaussianRasterizer
    from splatloc_amd.camera import PinholeCamera
    from splatloc_amd.optim import Adam as FusedAdam
    from splatloc_amd.synthetic import WORKLOADS, make_workload
    from splatloc_amd.training import color_refinement_step
    wl = WORKLOADS[args.workload]
    sc = make_workload(args.workload)
    P0, W, H, C = wl["P"], wl["W"], wl["H"], wl["C"]
    E = max(C - 3, 1)
    inv_sig = lambda p: torch.log(p / (1 - p))  # noqa: E731
    NAMES = ("xyz", "f_dc", "f_rest", "opacity", "marker", "kp_score", "scaling", "rotation")
    ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
            "marker": "_marker", "kp_score": "_kp_score", "scaling": "_scaling", "rotation": "_rotation"}
    LR = {"xyz": 1.6e-4 * 6.0, "f_dc": 2.5e-3, "f_rest": 2.5e-3 / 20, "opacity": 5e-2, "marker": 5e-2, "kp_score": 5e-2,
          "scaling": 1e-3 * 6.0, "rotation": 1e-3}

    def make_model(adam_cls, **adam_kw):
        g = torch.Generator().manual_seed(11)
        par = lambda t: torch.nn.Parameter(t.to(dev).contiguous().requires_grad_(True))  # noqa: E731
        pc = types.SimpleNamespace(
            _xyz=par(sc.means3D.clone()),
            _features_dc=par(((sc.features[:, :3] - 0.5) / 0.28209479177387814)[:, None, :].contiguous()),
            _features_rest=par(torch.zeros(P0, 0, 3)), _opacity=par(inv_sig(sc.opacities.clamp(1e-4, 1 - 1e-4))),
            _marker=par((torch.rand(P0, 1, generator=g) < 0.05).float() * torch.rand(P0, 1, generator=g) * 0.9),
            _kp_score=par(torch.rand(P0, E, generator=g)), _scaling=par(torch.log(sc.scales)),
            _rotation=par(sc.rotations.clone()), active_sh_degree=0, max_sh_degree=0,
            lr_init=1.6e-4 * 6.0, lr_final=1.6e-6 * 6.0, lr_delay_mult=0.01, max_steps=30000)
        pc.optimizer = adam_cls([{"params": [getattr(pc, ATTR[k])], "lr": LR[k], "name": k} for k in NAMES], lr=0.0,
                                eps=1e-15, **adam_kw)
        pc.max_radii2D = torch.zeros(P0, device=dev)
        return pc

    g = torch.Generator().manual_seed(12)
    views = []
    for k in range(8):
        ang = torch.tensor(0.02 * (k - 4))
        R = torch.tensor([[torch.cos(ang), 0, torch.sin(ang)], [0, 1, 0], [-torch.sin(ang), 0, torch.cos(ang)]])
        cam = PinholeCamera(W, H, W / 2.0, W / 2.0, (W - 1) / 2.0, (H - 1) / 2.0, R, torch.tensor([0.01 * k, 0.0, 0.0])).to(dev)
        cam.original_image = torch.rand(3, H, W, generator=g).to(dev)
        views.append(cam)
    bg = torch.zeros(3, device=dev)
    pipe = types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False)
    win = torch.tensor([math.exp(-((x - 5) ** 2) / 4.5) for x in range(11)])
    win = (win / win.sum()).to(dev)
    win2d = (win[:, None] @ win[None, :]).expand(3, 1, 11, 11).contiguous()

    def ssim_torch(a_, b_):   # loss_utils.py:72-102
        conv = lambda t: torch.nn.functional.conv2d(t, win2d, padding=5, groups=3)  # noqa: E731
        mu1, mu2 = conv(a_), conv(b_)
        s1, s2, s12 = conv(a_ * a_) - mu1 * mu1, conv(b_ * b_) - mu2 * mu2, conv(a_ * b_) - mu1 * mu2
        return (((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))).mean()

    def composed_step(pc, cam, it):   # train_gaussians.py:275-297 with the reference's torch ops around the per-view rasterizer
        rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg, 1.0,
                                           cam.world_view_transform, cam.full_proj_transform, 0, cam.camera_center,
                                           False, False)
        m2 = torch.zeros_like(pc._xyz, requires_grad=True) + 0
        feats = torch.cat((pc._features_dc, pc._features_rest), dim=1)
        rgb = torch.clamp_min(0.28209479177387814 * feats.transpose(1, 2).view(-1, 3, 1)[..., 0] + 0.5, 0.0)
        img, depth, alpha, radii = GaussianRasterizer(raster_settings=rs)(
            means3D=pc._xyz, means2D=m2, shs=None, colors_precomp=torch.cat((rgb, pc._kp_score), dim=1),
            opacities=torch.sigmoid(pc._opacity), scales=torch.exp(pc._scaling),
            rotations=torch.nn.functional.normalize(pc._rotation), cov3D_precomp=None)
        image, vis = img[:3], radii > 0
        gt = cam.original_image
        loss = 0.8 * torch.abs(image - gt).mean() + 0.2 * (1.0 - ssim_torch(image[None], gt[None]))
        loss.backward()
        key_mask = pc._marker.detach().squeeze() > 0.005
        pc._xyz.grad[key_mask] = 0
        with torch.no_grad():
            pc.max_radii2D[vis] = torch.max(pc.max_radii2D[vis], radii[vis].float())
            pc.optimizer.step()
            pc.optimizer.zero_grad(set_to_none=True)

    def time_it(fused):
        pc = make_model(FusedAdam) if fused else make_model(torch.optim.Adam, fused=True)
        for it in range(args.warmup):
            (color_refinement_step(views[it % 8], pc, pipe, bg, 0.2, it + 1) if fused else composed_step(pc, views[it % 8], it + 1))
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for it in range(args.steps):
            (color_refinement_step(views[it % 8], pc, pipe, bg, 0.2, it + 1) if fused else composed_step(pc, views[it % 8], it + 1))
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / args.steps * 1e3

    ms_f = time_it(True)
    ms_c = time_it(False)
    print(json.dumps({
        "metric": "SplatLoc.color_refinement iterations/s (1 view/iteration; secondary figure, NOT the BASELINE metric)",
        "value": round(1e3 / ms_f, 2), "unit": "iterations/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_f, 4), "higher_is_better": True, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: P={P0}, {W}x{H}, C={3 + E} ([rgb | kp_score]); per iteration: render, "
                               "0.8 L1 + 0.2 (1 - SSIM) on RGB, backward (3 colour channels, no depth / alpha terms), key gate, "
                               "max_radii2D, fused Adam over 8 groups, lr schedule; 8 different cameras in rotation"},
        "scene_of_26000_iterations_s": round(26000 * ms_f / 1e3, 1),
        "torch_front_end_loss_adam_same_rasterizer": {"ms_per_step": round(ms_c, 4), "speedup": round(ms_c / ms_f, 3)},
    }), flush=True)


def bench_scene(args, dev, rank=0, world=1):
    """--stage scene: the reference's WHOLE reconstruction schedule as one run (train_gaussians.py:310-355 `do_recon`;
    splatloc_amd.scene.do_recon): `--keyframes` synthetic RGB-D key-frames at the reference's frame size (640x480, Replica
    intrinsics) x (extend_from_pcd_seq + 10 map iterations, densify_and_prune every 150 / offset 50), then `--refine`
    colour-refinement iterations (the reference: 26 000), save_ply, and the forward-only eval_rendering loop with device
    PSNR / SSIM.  P grows from zero.  Reports wall time per phase next to the figure extrapolated from the per-iteration
    stage benches.  Secondary figure, not the BASELINE metric.

    `--gpus N` (BASELINE config 5), two forms:
      --replicas   N independent scenes (seed = rank), one per GPU, NO collective on the data path — what
                   /root/reference/replica.sh:1-6 does one scene after the other; value = N scenes / the slowest rank's time;
      (default)    ONE scene reconstructed frame-parallel: the views of every map window dealt to the ranks, two collectives
                   per step (`--reduce ring | rs_ag`), the one-view refinement run redundantly and re-broadcast once."""
    import tempfile
    import types
    from splatloc_amd.evaluation import eval_rendering
    from splatloc_amd.ply import save_ply
    from splatloc_amd.scene import DEFAULT_CONFIG, SceneModel, do_recon, synthetic_keyframes
    K, W, H = args.keyframes, 640, 480
    replicas = bool(args.replicas) and DIST_ON
    scene_seed = rank if replicas else 0
    t0 = time.perf_counter()
    frames, _ = synthetic_keyframes(K, W, H, P_truth=args.truth, seed=scene_seed, device=dev)
    torch.cuda.synchronize(dev)
    t_data = time.perf_counter() - t0
    pipe = types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False)
    bg = torch.zeros(3, device=dev)
    model = SceneModel(DEFAULT_CONFIG, dev)
    torch.cuda.reset_peak_memory_stats(dev)
    if DIST_ON:
        dist.barrier()
        torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    grp = None if (replicas or not DIST_ON) else dist.group.WORLD
    stats = do_recon(model, frames, pipe, bg, DEFAULT_CONFIG, refine_iterations=args.refine, seed=scene_seed,
                     batched=not args.no_window, group=grp, distributed=not replicas)
    t_recon = time.perf_counter() - t1
    with tempfile.TemporaryDirectory() as td:
        t2 = time.perf_counter()
        save_ply(model, os.path.join(td, "point_cloud.ply"))
        t_ply = time.perf_counter() - t2
        ply_bytes = os.path.getsize(os.path.join(td, "point_cloud.ply"))
    t3 = time.perf_counter()
    ev = eval_rendering(frames, model, [f.original_image for f in frames], pipe, bg, window=5)
    torch.cuda.synchronize(dev)
    t_eval = time.perf_counter() - t3
    map_ms = 1e3 * stats["map_seconds"] / max(stats["map_iterations"], 1)
    ref_ms = 1e3 * stats["refine_seconds"] / max(stats["refine_iterations"], 1)
    per_rank = [{"rank": rank, "recon_s": round(t_recon, 3), "map_s": round(stats["map_seconds"], 3),
                 "refine_s": round(stats["refine_seconds"], 3), "rows_final": stats["rows_final"],
                 "mean_psnr": round(ev["mean_psnr"], 3), "peak_memory_GB": round(stats["peak_memory_bytes"] / 2 ** 30, 3)}]
    if DIST_ON:
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])       # reporting only (control plane): after the timed schedule
        per_rank = gathered
    if rank != 0:
        return
    t_max = max(r["recon_s"] for r in per_rank)
    n_scenes = world if replicas else 1
    mode = "replicas" if replicas else ("frame-parallel" if DIST_ON else "single")
    print(json.dumps({
        "metric": "SplatLoc.do_recon scenes/hour on synthetic key-frames (whole schedule; secondary figure, NOT the BASELINE metric)",
        "value": round(3600.0 * n_scenes / t_max, 3), "unit": "scenes/h", "n_gpus": world, "steps": 1, "warmup": 0,
        "ms_per_step": round(1e3 * t_max, 1), "higher_is_better": True, "dtype": "f32", "data": "synthetic",
        "scaling": "weak" if replicas else "strong",
        "config": {"workload": f"{K} key-frames {W}x{H} (Replica intrinsics) of a {args.truth}-Gaussian synthetic room; per key-frame "
                               "extend_from_pcd_seq + 10 map iterations (window 5, densify every 150 offset 50, reset every 2001); "
                               f"{args.refine} color_refinement iterations; save_ply; eval_rendering",
                   "launch_mode": "per-view calls" if args.no_window else "window-batched",
                   "multi_gpu_mode": mode, "scenes": n_scenes,
                   "collectives_on_the_data_path": 0 if (replicas or not DIST_ON) else "2 payload + 1 header per map step, 1 broadcast after the refinement",
                   "reduce": None if (replicas or not DIST_ON) else args.reduce,
                   "dist_backend": None if not DIST_ON else dist.get_backend(),
                   "dist_env": dist_env_report() if DIST_ON else None},
        "per_rank": per_rank,
        "seconds": {"synthetic_keyframes": round(t_data, 2), "map_phase": round(stats["map_seconds"], 2),
                    "refine_phase": round(stats["refine_seconds"], 2), "save_ply": round(t_ply, 3), "eval_rendering": round(t_eval, 3)},
        "map_ms_per_iteration": round(map_ms, 3), "refine_ms_per_iteration": round(ref_ms, 4),
        "refine_26000_iterations_s_extrapolated_from_this_run": round(26000 * ref_ms / 1e3, 1),
        "render_paths_of_the_map_steps": stats.get("render_paths"),
        "rows_after_keyframe": stats["rows_after_keyframe"], "rows_final": stats["rows_final"],
        "densifications": stats["densify_rows"], "peak_memory_GB": round(stats["peak_memory_bytes"] / 2 ** 30, 3),
        "ply_bytes": ply_bytes,
        "eval": {"mean_psnr": round(ev["mean_psnr"], 3), "mean_ssim": round(ev["mean_ssim"], 4), "frames": ev["frames"],
                 "frames_per_s": round(ev["frames"] / t_eval, 1)},
    }), flush=True)


def bench_pose_refine(args, dev):
    """--stage pose_refine: BASELINE config 4 as it is worded — "feature raster + pose refinement" — at full size: 500k
    Gaussians (S2's scene, SplatLoc's [rgb | kp] layout, C = 4), one 640x480 query frame at the 12-Scenes intrinsics
    (configs/scenes12/base_config.yaml:17-27: fx = fy = 572, cx = 320, cy = 240), the pose perturbed by ~1.5 degrees / 5 cm.
    One iteration = render, L1 colour + 0.2 L1 depth, backward with dL/dviewmatrix / dL/dprojmatrix, Adam on (axis-angle,
    translation), next camera tensors (splatloc_amd.pose.refine_pose; utils/optimization_utils.py:31-42 is the
    parameterisation).  Reported: iterations/s of the graph-free loop (csrc/pose.hip) and of round 4's autograd loop, and the
    live GPU idle of the graph-free loop (wall time of an un-instrumented region vs the kernel time of the same loop under
    torch.profiler, like tools/refine_idle.py).  The reference has no such loop: a build extension, secondary figure."""
    from splatloc_amd import GaussianRasterizationSettings, GaussianRasterizer, pose
    from splatloc_amd.camera import PinholeCamera
    from splatloc_amd.synthetic import make_workload
    sc = make_workload("S2-ref-layout").to(dev)
    P = int(sc.means3D.shape[0])
    W, H = 640, 480
    cam = PinholeCamera(W, H, 572.0, 572.0, 320.0, 240.0)
    cam.to(dev)
    W2C_true = pose.at_to_transform_matrix(torch.tensor([[0.015, -0.02, 0.008]], device=dev),
                                           torch.tensor([[0.03, -0.02, 0.04]], device=dev))[0]
    with torch.no_grad():
        view, proj, campos = pose.camera_tensors(W2C_true, cam.projection_matrix)
        rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, sc.bg, 1.0, view, proj, 0, campos, False, False)
        tgt_c, tgt_d, _, _ = GaussianRasterizer(raster_settings=rs)(
            means3D=sc.means3D, means2D=torch.zeros_like(sc.means3D), shs=None, colors_precomp=sc.features,
            opacities=sc.opacities, scales=sc.scales, rotations=sc.rotations, cov3D_precomp=None)
    g = dict(means3D=sc.means3D, colors=sc.features, opacities=sc.opacities, scales=sc.scales, rotations=sc.rotations)
    N = max(args.steps, 20) * 5
    W2C0 = torch.eye(4, device=dev)

    def run(graph_free, n):
        return pose.refine_pose((tgt_c, tgt_d), g, cam, W2C0, iterations=n, background=sc.bg, graph_free=graph_free)

    out = {}
    for name, gf in (("graph_free", True), ("autograd_loop", False)):
        run(gf, 20)
        torch.cuda.synchronize(dev)
        walls = []
        for _ in range(max(args.repeats, 1)):
            t0 = time.perf_counter()
            W2C, hist = run(gf, N)
            torch.cuda.synchronize(dev)
            walls.append((time.perf_counter() - t0) / N)
        walls.sort()
        out[name] = {"ms_per_iteration": round(1e3 * walls[len(walls) // 2], 4), "iterations_per_s": round(1.0 / walls[len(walls) // 2], 1),
                     "ms_all_regions": [round(1e3 * w, 4) for w in walls],
                     "loss_first_last": [round(float(hist[0]), 6), round(float(hist[-1]), 6)]}
    # live idle of the graph-free loop
    from torch.profiler import ProfilerActivity, profile
    M = 60
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        run(True, M)
        torch.cuda.synchronize(dev)
    rows = []
    for e in prof.key_averages():
        dt = getattr(e, "device_time_total", None)
        if dt is None:
            dt = getattr(e, "cuda_time_total", 0)
        if dt and e.count:
            rows.append((e.key[:70], e.count / M, dt / M))
    rows.sort(key=lambda r: -r[2])
    busy_us = sum(r[2] for r in rows)
    wall_us = 1e3 * out["graph_free"]["ms_per_iteration"]

    def pose_err(M4):
        dR = M4[:3, :3] @ W2C_true[:3, :3].T
        return float(torch.acos(((torch.trace(dR) - 1) / 2).clamp(-1, 1))) + float((M4[:3, 3] - W2C_true[:3, 3]).norm())

    Wf, _ = run(True, 300)
    print(json.dumps({
        "metric": "pose-refinement iterations/s (render + L1 RGB-D + backward with pose gradients + Adam on 6 parameters; build extension, NOT the BASELINE metric)",
        "value": out["graph_free"]["iterations_per_s"], "unit": "iterations/s", "n_gpus": 1, "steps": N, "warmup": 20,
        "ms_per_step": out["graph_free"]["ms_per_iteration"], "higher_is_better": True, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"P={P} Gaussians (S2's scene), one {W}x{H} query frame, fx = fy = 572 (12-Scenes), C = 4 ([rgb | kp]); "
                               "pose = T(axis-angle, translation) @ W2C_init, perturbed by ~1.5 deg / 5 cm"},
        "graph_free": out["graph_free"], "autograd_loop": out["autograd_loop"],
        "speedup_over_autograd_loop": round(out["autograd_loop"]["ms_per_iteration"] / out["graph_free"]["ms_per_iteration"], 3),
        "idle": {"wall_us_per_iteration": round(wall_us, 1), "gpu_busy_us_per_iteration_torch_profiler": round(busy_us, 1),
                 "idle_us_per_iteration": round(wall_us - busy_us, 1), "kernels_per_iteration": round(sum(r[1] for r in rows), 1)},
        "kernel_table_us_per_iteration": [{"kernel": k, "launches": round(c, 2), "us": round(u, 1)} for k, c, u in rows[:24]],
        "pose_error_rad_plus_m": {"start": round(pose_err(W2C0), 5), "after_300_iterations": round(pose_err(Wf), 5)},
    }), flush=True)


def bench_eval_rendering(args, dev):
    """--stage eval_rendering: the reference's eval loop (utils/eval_utils.py:22-72; BASELINE config 4's stand-in)
    at the 12-Scenes intrinsics (configs/scenes12/base_config.yaml:17-27: 640x480, fx = fy = 572, cx = 320, cy = 240):
    forward-only renders under no_grad + clamp + PSNR (masked) + SSIM per frame, as windows of 5 frames
    (splatloc_amd.evaluation.eval_rendering) and frame by frame (the reference's loop), on S2's 500k Gaussians with
    SplatLoc's [rgb | kp_score] layout.  Secondary figure, not the BASELINE metric."""
    import types
    from splatloc_amd.camera import PinholeCamera
    from splatloc_amd.evaluation import eval_rendering
    from splatloc_amd.synthetic import make_workload
    sc = make_workload("S2-ref-layout")
    P = int(sc.means3D.shape[0])
    W, H, fx, cx, cy = 640, 480, 572.0, 320.0, 240.0
    par = lambda t: torch.nn.Parameter(t.to(dev).contiguous())  # noqa: E731
    pc = types.SimpleNamespace(
        _xyz=par(sc.means3D.clone()), _features_dc=par(((sc.features[:, :3] - 0.5) / 0.28209479177387814)[:, None, :].contiguous()),
        _features_rest=par(torch.zeros(P, 0, 3)), _opacity=par(torch.logit(sc.opacities.clamp(1e-4, 1 - 1e-4))),
        _kp_score=par(sc.features[:, 3:4].clone()), _scaling=par(torch.log(sc.scales)), _rotation=par(sc.rotations.clone()),
        active_sh_degree=0, max_sh_degree=0)
    g = torch.Generator().manual_seed(21)
    frames, gts = [], []
    for k in range(40):
        ang = torch.tensor(0.015 * (k - 20))
        R = torch.tensor([[torch.cos(ang), 0, torch.sin(ang)], [0, 1, 0], [-torch.sin(ang), 0, torch.cos(ang)]])
        frames.append(PinholeCamera(W, H, fx, fx, cx, cy, R, torch.tensor([0.01 * k, 0.0, 0.0])).to(dev))
        gts.append(torch.rand(3, H, W, generator=g).to(dev))
    pipe = types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False)
    bg = torch.zeros(3, device=dev)

    def time_it(window):
        for _ in range(max(args.warmup, 1)):
            eval_rendering(frames, pc, gts, pipe, bg, window=window)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = eval_rendering(frames, pc, gts, pipe, bg, window=window)
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / args.steps, out

    s5, out5 = time_it(5)
    s1, out1 = time_it(1)
    assert out5["psnr"] == out1["psnr"] and out5["ssim"] == out1["ssim"]       # bit-identical images either way
    print(json.dumps({
        "metric": "eval_rendering frames/s (forward only + PSNR + SSIM; secondary figure, NOT the BASELINE metric)",
        "value": round(len(frames) / s5, 1), "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * s5, 3), "higher_is_better": True, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"P={P} Gaussians, {W}x{H}, fx = fy = {fx} (12-Scenes), C = 4 ([rgb | kp_score]); 40 frames per pass; "
                               "render under no_grad + clamp + masked PSNR + SSIM per frame; windows of 5 frames"},
        "frame_by_frame_loop": {"frames_per_s": round(len(frames) / s1, 1), "ms_per_pass": round(1e3 * s1, 3)},
        "mean_psnr": round(out5["mean_psnr"], 4), "mean_ssim": round(out5["mean_ssim"], 5),
    }), flush=True)


def start_gpu_sampler(hz: float = 40.0):
    """tools/gpu_sampler.py as a child process (amdsmi only, no HIP); returns (Popen, path) or None when it cannot start."""
    import subprocess
    import tempfile
    path = os.path.join(tempfile.gettempdir(), f"splatloc_gpu_samples_{os.getpid()}.json")
    try:
        p = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "gpu_sampler.py"), path, str(hz)], stdin=subprocess.PIPE,
                             stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=ROOT)
    except OSError:
        return None
    return p, path


def stop_gpu_sampler(sampler, spans):
    """Close the sampler's stdin (it writes its samples and exits) and reduce the samples inside `spans`."""
    p, path = sampler
    try:
        p.stdin.close()
        p.wait(timeout=20)
        with open(path) as f:
            data = json.load(f)
        os.remove(path)
    except Exception as ex:  # noqa: BLE001
        try:
            p.kill()
        except Exception:  # noqa: BLE001
            pass
        return {"available": False, "why": f"sampler: {ex!r}"[:160]}
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gpu_sampler import summarize
    return summarize(data, spans)


DIST_ON = False      # a process group exists (N > 1 ranks, or --force-process-group's group of one)


def dist_env_report() -> dict:
    """What the collectives of this run ran on: backend, RCCL version, the IPC mode variable RCCL needs on this pool."""
    rep = {"backend": dist.get_backend() if dist.is_initialized() else None, "world_size": dist.get_world_size() if dist.is_initialized() else 1,
           "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
           "NCCL_DEBUG": os.environ.get("NCCL_DEBUG"), "torch": torch.__version__, "hip": getattr(torch.version, "hip", None)}
    try:
        rep["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception as e:  # noqa: BLE001
        rep["rccl_version"] = f"unavailable ({type(e).__name__})"
    return rep


def count_gpus_without_hip() -> int:
    """GPUs of this node as the KFD driver lists them (/sys/class/kfd/kfd/topology/nodes/*/properties: a node with
    simd_count > 0 is a GPU), narrowed by ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES — read without
    loading the HIP runtime, so the process that only LAUNCHES the ranks provably never initialises a GPU.
    Falls back to torch.cuda.device_count() (which on this image does not initialise HIP either) when sysfs is unreadable."""
    import glob
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    n, seen = 0, False
    for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            with open(path) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            continue
        seen = True
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    return n if seen else torch.cuda.device_count()


def launch_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks (one process per GPU) as a FRESH child process —
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py <same args>` — and return
    its exit code; rank 0's JSON line goes straight to the inherited stdout.  Called before anything touches the GPU (this
    process never initialises HIP: the GPUs are counted from the KFD topology in sysfs, `count_gpus_without_hip`), and it
    starts a child, it never re-execs.  SPLATLOC_DIST_BACKEND=gloo lets a 1-GPU box drive the N-rank path (tests); with RCCL ("nccl", the default)
    N ranks need N GPUs and anything less is refused instead of silently measuring fewer."""
    import socket
    import subprocess
    backend = os.environ.get("SPLATLOC_DIST_BACKEND", "nccl")
    ndev = count_gpus_without_hip()
    if backend == "nccl" and ndev < n:
        print(f"bench.py --gpus {n}: this node shows {ndev} GPU(s); RCCL needs one GPU per rank "
              "(SPLATLOC_DIST_BACKEND=gloo runs the N-rank plumbing on fewer GPUs, for tests only)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    sys.stdout.flush()
    return subprocess.run(cmd, env=env, cwd=ROOT, stdout=sys.stdout).returncode      # (the ranks' fd 1 = the REAL stdout: main() moved ours)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="S2")
    ap.add_argument("--views", type=int, default=5,
                    help="frames per optimisation step (SplatLoc.map's window_size = 5); gradients accumulate")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = every rank renders its own window; strong = one window dealt to the ranks")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the views of a window are spread over (1 = the reference's serial loop)")
    ap.add_argument("--densify-every", type=int, default=10,
                    help="--stage map_step: densify_and_prune every N steps (the reference: 150, offset 50)")
    ap.add_argument("--no-window", action="store_true",
                    help="render the window as a loop of per-view GaussianRasterizer calls (the reference's loop) instead of "
                         "ONE window-batched launch sequence")
    ap.add_argument("--no-multi-stream", action="store_true", help="skip the secondary legs (per-view loop, multi-stream) (profiling runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fwd-only", action="store_true", help="debug: time the forward only (not the metric)")
    ap.add_argument("--keyframes", type=int, default=60, help="--stage scene: key-frames of the synthetic scene")
    ap.add_argument("--refine", type=int, default=26000, help="--stage scene: color_refinement iterations (the reference: 26000)")
    ap.add_argument("--replicas", action="store_true",
                    help="--stage scene --gpus N: N INDEPENDENT scenes, one per GPU, no collective on the data path (what "
                         "/root/reference/replica.sh runs one after the other); without it the one scene is reconstructed "
                         "frame-parallel by the N ranks")
    ap.add_argument("--reduce", default="ring", choices=["ring", "rs_ag"],
                    help="N > 1: the SUM exchange of a step as one all-reduce (ring) or as reduce-scatter + all-gather (rs_ag)")
    ap.add_argument("--force-process-group", action="store_true",
                    help="--gpus 1: create a world-size-1 RCCL ('nccl') process group with device_id= and issue EVERY collective of the "
                         "N > 1 path on it (in-place span SUM, MAX, header, barrier; --reduce rs_ag: the aliased reduce-scatter + "
                         "all-gather pair) — first contact with RCCL on a one-GPU box; values are unchanged by construction")
    ap.add_argument("--no-telemetry", action="store_true",
                    help="do not start tools/gpu_sampler.py (amdsmi activity / clock / power samples across the timed regions)")
    ap.add_argument("--truth", type=int, default=200_000, help="--stage scene: Gaussians of the synthetic ground-truth room")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed region of K steps is repeated this many times; value = the MEDIAN region (min / max reported)")
    ap.add_argument("--stage", default="raster",
                    choices=["raster", "activations", "loss", "map_step", "refine_step", "scene", "eval_rendering", "pose_refine"],
                    help="raster = the BASELINE metric (default); activations = the fused front-end stage alone")
    args = ap.parse_args()

    # stdout carries ONE JSON line.  Native libraries write banners to file descriptor 1 (RCCL under NCCL_DEBUG=VERSION: five lines per
    # process group; gloo: "[Gloo] Rank 0 is connected to ..."): Python's own stdout moves to a duplicate of the real one and fd 1 is
    # pointed at stderr, so that whatever C code prints lands beside the warnings, not in front of the line the driver parses.
    sys.stdout.flush()
    sys.stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: become one (a fresh child process per rank; nothing here has touched the GPU yet)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; "
                         "the JSON line's n_gpus must be what was asked for")
    # telemetry: a separate process samples the amdsmi metrics table (busy %, shader clock, socket power) across the run; it is
    # started HERE, before anything in this process touches the GPU, and it never loads HIP itself
    sampler = None
    if rank == 0 and args.stage == "raster" and not args.no_telemetry:
        sampler = start_gpu_sampler()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)   # > 1 rank per GPU only in the single-GPU plumbing test
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    global DIST_ON
    DIST_ON = world > 1 or args.force_process_group
    if DIST_ON:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1 and "MASTER_PORT" not in os.environ:     # --force-process-group without a launcher: a group of one
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.force_process_group:
            from splatloc_amd import frame_parallel as _fp
            _fp.FORCE_COLLECTIVES = True
        # "nccl" is RCCL on ROCm (xGMI); SPLATLOC_DIST_BACKEND=gloo lets tests drive this exact
        # code path with several ranks on one GPU.
        backend = os.environ.get("SPLATLOC_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        from splatloc_amd import frame_parallel as _fp2
        _fp2.prepare()          # the length check's gloo twin group: created here, not inside the first step

    from splatloc_amd import training as _training
    _training.REDUCE_MODE = args.reduce
    if args.stage == "scene" and DIST_ON:
        bench_scene(args, dev, rank, world)
        dist.barrier()
        dist.destroy_process_group()
        return
    if args.stage in ("activations", "loss", "map_step", "refine_step", "scene", "eval_rendering", "pose_refine"):
        if rank == 0:
            {"activations": bench_activations, "loss": bench_loss, "map_step": bench_map_step,
             "refine_step": bench_refine_step, "scene": bench_scene, "eval_rendering": bench_eval_rendering,
             "pose_refine": bench_pose_refine}[args.stage](args, dev)
        if DIST_ON:
            dist.barrier()
            dist.destroy_process_group()
        return

    from splatloc_amd import GaussianRasterizationSettings, GaussianRasterizer, _native, rasterize_window
    from splatloc_amd.camera import PinholeCamera
    from splatloc_amd.densify import add_densification_stats, add_densification_stats_window
    from splatloc_amd.frame_parallel import reduce_step, shard_views
    from splatloc_amd.rasterizer import window_grad_span
    from splatloc_amd.synthetic import WORKLOADS, make_workload

    wl = WORKLOADS[args.workload]
    sc = make_workload(args.workload).to(dev)
    P, W, H, C = wl["P"], wl["W"], wl["H"], wl["C"]
    leaf = lambda t: t.clone().requires_grad_(True)  # noqa: E731
    means3D, colors, opac = leaf(sc.means3D), leaf(sc.features), leaf(sc.opacities)
    scales, rots = leaf(sc.scales), leaf(sc.rotations)
    params = [means3D, colors, opac, scales, rots]

    # The window: `--views` DIFFERENT cameras (small rotations / translations about the scene's camera, as
    # consecutive key-frames are), different on every rank in weak mode.  Nothing of one frame (lists,
    # payload, images) can be found in a cache by the next.  Strong mode deals ONE window to the ranks.
    def make_view(j):
        ang = torch.tensor(0.02 * ((j % 5) - 2) + 0.0037 * (j // 5))
        Rm = torch.tensor([[torch.cos(ang), 0, torch.sin(ang)], [0, 1, 0], [-torch.sin(ang), 0, torch.cos(ang)]])
        cam = PinholeCamera(W, H, W / 2.0, W / 2.0, (W - 1) / 2.0, (H - 1) / 2.0, Rm,
                            torch.tensor([0.01 * (j % 5) + 0.002 * (j // 5), 0.0, 0.0])).to(dev)
        rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, sc.bg, 1.0, cam.world_view_transform,
                                           cam.full_proj_transform, 0, cam.camera_center, False, False)
        # a different dL/dout per view: the seeded gradient images, rolled
        g = tuple(torch.roll(t, shifts=37 * j, dims=-1).contiguous() for t in (sc.dL_dcolor, sc.dL_ddepth, sc.dL_dalpha))
        return GaussianRasterizer(raster_settings=rs), g, rs

    if args.scaling == "strong":
        my_ids = shard_views(list(range(args.views)), rank, world)
    else:
        my_ids = [rank * args.views + k for k in range(args.views)]
    views = [make_view(j) for j in my_ids]
    frames_per_step = args.views if args.scaling == "strong" else world * args.views
    accum = torch.zeros(P, 1, device=dev)
    denom = torch.zeros(P, 1, device=dev)
    max_radii = torch.zeros(P, device=dev)
    info = {"R": [], "V": []}

    side = [torch.cuda.Stream(device=dev) for _ in range(args.streams)] if args.streams > 1 else []
    mode = {"window": not args.no_window and args.streams <= 1}

    def window_step(record):
        # the whole window as ONE launch sequence: forward of the V views, ONE backward that sums their parameter
        # gradients in-kernel, then the per-view densification statistics (train_gaussians.py:238-245)
        block = torch.zeros((len(views),) + tuple(means3D.shape), dtype=means3D.dtype, device=means3D.device)
        carriers = [block[k].requires_grad_(True) for k in range(len(views))]       # render(): screenspace_points per view (one zero fill)
        span = [] if DIST_ON else None    # N > 1: the backward's gradient allocation ends with a [2, P] tail for the statistics increments
        outs = rasterize_window([rs for _, _, rs in views], means3D, carriers, colors, opac, scales=scales, rotations=rots,
                                grad_span=span)
        if record:
            for k in range(0, len(outs), 8):
                info["R"] += [int(r) for r in outs[k][0].grad_fn.R]
            info["V"] += [int((o[3] > 0).sum().item()) for o in outs]
        if not args.fwd_only:
            torch.autograd.backward([t for o in outs for t in o[:3]], [g for _, gs, _ in views for g in gs])
            if span is not None and len(span) == 1:
                # the step's increments of xyz_gradient_accum / denom go into the tail of the gradient allocation: they are
                # summed over the ranks by the SAME in-place all-reduce as the gradients (frame_parallel.reduce_step)
                inc = span[0]["tail"]
                add_densification_stats_window([m2.grad for m2 in carriers], [o[3] for o in outs], inc[0], inc[1], max_radii)
                return inc
            add_densification_stats_window([m2.grad for m2 in carriers], [o[3] for o in outs], accum, denom, max_radii)
        return None

    def one_view(rast, g_out, record):
        means2D = torch.zeros_like(means3D, requires_grad=True)   # per-view grad carrier (render(): screenspace_points)
        color, depth, alpha, radii = rast(means3D=means3D, means2D=means2D, shs=None, colors_precomp=colors,
                                          opacities=opac, scales=scales, rotations=rots, cov3D_precomp=None)
        if record:
            info["R"].append(int(color.grad_fn.num_rendered))
            info["V"].append(int((radii > 0).sum().item()))
        if not args.fwd_only:
            torch.autograd.backward((color, depth, alpha), g_out)
            # the step's densification statistics (train_gaussians.py:238-245): the same per-frame work at every N
            add_densification_stats(means2D.grad, radii, accum, denom, max_radii)

    def step(record=False):
        for p in params:
            p.grad = None
        inc = None
        if DIST_ON and not (mode["window"] and views):
            accum.zero_()      # the per-view paths accumulate the step's increments here (summed over the ranks below)
            denom.zero_()
        if mode["window"] and views:
            inc = window_step(record)
        elif side:
            # the views of a window are independent until their gradients are summed: view j runs on HIP stream
            # j % K, so the latency-bound kernels of a small frame overlap with another view's (autograd runs
            # each backward on its forward's stream and orders the accumulation into .grad)
            main = torch.cuda.current_stream(dev)
            for st in side:
                st.wait_stream(main)
            for j, (rast, g_out, _) in enumerate(views):
                with torch.cuda.stream(side[j % len(side)]):
                    one_view(rast, g_out, record)
            for st in side:
                main.wait_stream(st)
        else:
            for rast, g_out, _ in views:   # every frame one forward + one backward, parameter gradients accumulate
                one_view(rast, g_out, record)
        if not args.fwd_only and DIST_ON:
            # TWO collectives per step (frame_parallel.reduce_step): ONE SUM all-reduce over [accumulated parameter gradients |
            # increments of xyz_gradient_accum, denom] — in place in the backward's own allocation on the window path — and
            # ONE MAX all-reduce of max_radii2D, so that every replica would take the same optimizer step and densify identically
            if not views:
                # a rank without views (strong mode, more ranks than views) contributes zeros in EXACTLY the layout the
                # other ranks' backward produced
                z = window_grad_span(P, C, dev, have_scales=True, have_cov=False, tail=True, zero=True)
                for p, k in zip(params, ("m3", "col", "op", "sca", "rot")):
                    p.grad = z[k]
                inc = z["tail"]
            if inc is not None:
                _, inc_out, red = reduce_step([p.grad for p in params], sum_extras=[inc[0], inc[1]], max_extras=[max_radii],
                                              mode=args.reduce)
                accum.add_(inc_out[0])
                denom.add_(inc_out[1])
            else:   # per-view calls (--no-window / --streams): autograd accumulated into the first view's allocation
                g_out, inc_out, red = reduce_step([p.grad for p in params], sum_extras=[accum, denom], max_extras=[max_radii],
                                                  mode=args.reduce)
                for p, g in zip(params, g_out):
                    p.grad = g
                accum.copy_(inc_out[0])
                denom.copy_(inc_out[1])
            info["reduce_path"] = red

    def barrier():
        if DIST_ON:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    step(record=True)
    # (1) untimed breakdown pass: every stage bracketed by HIP events.  An event record between
    #     two kernels idles the GPU for ~10 us, so this is NOT done inside the timed region.
    barrier()
    _native.timing_enable(True)
    _native.timing_collect()
    for _ in range(BREAKDOWN_STEPS):
        step()
    barrier()
    _native.timing_enable(False)
    stages = _native.timing_collect()
    dom = max(stages, key=lambda s: stages[s][0]) if views else "composite_bwd"
    # (2) timed region: only the dominant kernel is bracketed (roofline.achieved is measured live
    #     here, on the launch stream)
    _native.timing_select([dom])
    regions, regions_unix = [], []
    for _ in range(max(args.repeats, 1)):       # every region: barrier + synchronize, EXACTLY K steps, barrier + synchronize
        barrier()
        t0, u0 = time.perf_counter(), time.time()
        for _ in range(args.steps):
            step()
        barrier()
        regions.append(time.perf_counter() - t0)
        regions_unix.append([u0, time.time()])
    _native.timing_enable(False)
    stages[dom] = _native.timing_collect()[dom]
    if DIST_ON:       # MAX over the ranks, region by region
        t = torch.tensor(regions, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        regions = [float(x) for x in t.tolist()]
    elapsed = sorted(regions)[len(regions) // 2]     # value = the MEDIAN region
    telemetry = stop_gpu_sampler(sampler, regions_unix) if sampler is not None else None

    # Secondary figure (N = 1): the same step with the window's views spread over K HIP streams
    # (splatloc_amd.fused.render_window does this for the product path).  Not `value`: kernel durations
    # measured under overlap are not the dominant kernel's own, so the roofline stays on the serial run.
    multi_stream = []
    if world == 1 and not side and not args.fwd_only and len(views) > 1 and not args.no_multi_stream:
        was_window = mode["window"]
        mode["window"] = False
        for K in ((1, 2, 4) if was_window else (2, 4)):      # K = 1: the reference's loop of per-view calls
            side[:] = [torch.cuda.Stream(device=dev) for _ in range(K)] if K > 1 else []
            for _ in range(2):
                step()
            barrier()
            ts = time.perf_counter()
            for _ in range(args.steps):
                step()
            barrier()
            el = time.perf_counter() - ts
            multi_stream.append({"path": "per-view GaussianRasterizer calls", "streams": K,
                                 "value": round(frames_per_step * args.steps / el, 3),
                                 "ms_per_step": round(1e3 * el / args.steps, 4)})
        side[:] = []
        mode["window"] = was_window

    if rank == 0:
        Rs, Vs = info["R"], info["V"]
        R = int(round(sum(Rs) / max(len(Rs), 1)))
        V = int(round(sum(Vs) / max(len(Vs), 1)))
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        st_bytes, frame_bytes, n_pass = algorithmic_bytes(P, R, W, H, C, tiles)
        act_bytes = actual_bytes(P, V, R, W, H, C, tiles)
        ms_per_step = 1e3 * elapsed / args.steps
        value = frames_per_step * args.steps / elapsed
        # units per launch: a window-batched launch processes the frames of all the rank's views at once
        vpl = max(len(views), 1) if mode["window"] else 1
        per_stage = {}
        for s, (ms, cnt) in stages.items():
            if cnt:
                avg = ms / cnt
                per_stage[s] = {"avg_ms": round(avg, 4), "launches": int(cnt), "frames_per_launch": vpl,
                                "lineage_bytes": int(st_bytes[s]) * vpl, "actual_bytes": int(act_bytes[s]) * vpl,
                                "actual_GBps": round(act_bytes[s] * vpl / (avg * 1e-3) / 1e9, 1)}
        per_stage[dom]["measured"] = "live in the timed region"
        dom_avg = stages[dom][0] / max(stages[dom][1], 1)
        ach = round(st_bytes[dom] * vpl / (dom_avg * 1e-3) / 1e9, 1)     # roofline: SURVEY §8d algorithmic bytes / duration
        # Counter-derived figures are REPLAYS of the builder's rocprofv3 runs (profiles/*.json), not live
        # measurements: they are attached only when the stored run is this workload in this launch mode.
        prof = {}
        for name in ("traffic.json", "valu.json"):
            tpath = os.path.join(ROOT, "profiles", name)
            if os.path.exists(tpath):
                try:
                    j = json.load(open(tpath))
                    if j.get("_workload", "S2") == args.workload and int(j.get("_frames_per_launch", 1)) == vpl:
                        prof[name] = j
                except Exception:  # noqa: BLE001
                    pass
        traffic = prof.get("traffic.json", {}).get(dom)
        valu = prof.get("valu.json")
        roofline_valu = None
        clk = None
        import glob as _glob
        # the newest clock summary of a builder run (profiles/rNN_clocks.json, tools/clock_trace.py): ONE copy, no "clocks.json" twin
        cfiles = sorted(_glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_clocks.json")))
        cpath = cfiles[-1] if cfiles else ""
        if cpath:
            try:
                cj = json.load(open(cpath))
                a = cj.get("amdsmi_library_20hz") or {}
                res = a.get("residency_counters_first_last") or {}
                acc, ppt = res.get("accumulation_counter"), res.get("ppt_residency_acc")
                clk = {"achieved_sclk_mhz": (a.get("gfxclk_mhz_mean_over_xcds") or {}).get("median"),
                       "socket_power_w": (a.get("socket_power_w") or {}).get("median"), "power_cap_w": cj.get("power_cap_w"),
                       "ppt_limit_residency": (round((ppt[1] - ppt[0]) / max(acc[1] - acc[0], 1), 3) if acc and ppt else None)}
            except Exception:  # noqa: BLE001
                clk = None
        if valu:
            # the binding bound of the compositing kernels as a fraction: a wave64 VALU instruction occupies its
            # SIMD's issue port for 4 cycles (quad-cycle), 1024 SIMDs: issue fraction = wave-instr x 4 / (1024 x cycles)
            roofline_valu = {"source": "profiles/valu.json (builder rocprofv3 SQ-counter run, same workload and launch mode)",
                             "bound": "valu-issue", "kernels": {},
                             # measured clock of a >= 10-s loop of this workload (tools/clock_trace.py: amdsmi at ~20 Hz from a separate
                             # process; a replay of the newest profiles/rNN_clocks.json, not live): the SQ fractions below are fractions of ACHIEVED
                             # cycles (kernel cycles from GRBM_GUI_ACTIVE); `..._at_peak_clock` re-states them against 2.4 GHz
                             "clock": clk, "peak_sclk_mhz": 2400}
            for k, c in valu.items():
                if isinstance(c, dict) and c.get("kernel_cycles"):
                    cyc = c["kernel_cycles"]
                    issue = (c["valu_wave_instructions"] + c.get("mfma_wave_instructions", 0)) * 4
                    live_ms = dom_avg if k.startswith(dom + "_kernel") else None     # this run's measured duration of the dominant kernel
                    roofline_valu["kernels"][k] = {
                        "valu_issue_frac": round(issue / (1024 * cyc), 4),
                        "valu_issue_frac_at_peak_clock": (round(issue / (1024 * live_ms * 1e-3 * 2.4e9), 4) if live_ms else None),
                        "kernel_cycles_per_live_duration_mhz": (round(cyc / (live_ms * 1e-3) / 1e6, 1) if live_ms else None),
                        "valu_busy": c.get("valu_busy"), "mfma_busy": c.get("mfma_busy"),
                        "salu_per_valu": round(c.get("salu_wave_instructions", 0) / max(c["valu_wave_instructions"], 1), 3)}
        out = {
            "metric": "fwd+bwd frames/s @1080p, 500k Gaussians, 32 feat-ch; HBM GB/s vs roofline",
            "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "repeats": {"regions": len(regions), "steps_per_region": args.steps, "value_is": "median region",
                        "frames_per_s_min": round(frames_per_step * args.steps / max(regions), 3),
                        "frames_per_s_max": round(frames_per_step * args.steps / min(regions), 3),
                        "ms_per_step_all": [round(1e3 * r / args.steps, 4) for r in regions]},
            "timed_regions_unix": [[round(a, 3), round(b, 3)] for a, b in regions_unix],   # for tools/clock_trace.py
            # what the GPU did DURING the timed regions, from an independent source (amdsmi, sampled by a separate process)
            "gpu_busy_in_timed_regions": (telemetry or {}).get("busy_pct_mean"),
            "gpu_telemetry_in_timed_regions": telemetry,
            "config": {"workload": f"{args.workload}: P={P} Gaussians, {W}x{H}, C={C} channels "
                                   f"(3 RGB + {C - 3} feature) + depth + alpha, seed {wl['seed']}; "
                                   f"{args.views} different cameras per window",
                       "tile_instances_R_per_view": Rs, "visible_gaussians_V_per_view": Vs,
                       "tile_instances_R": R, "visible_gaussians_V": V,
                       "frames_per_step": frames_per_step, "views_on_rank0_per_step": len(views),
                       "hip_streams_per_window": max(args.streams, 1),
                       "launch_mode": ("window-batched: one launch sequence per window of views (splatraster_forward_window_* / "
                                       "splatraster_backward_window)") if mode["window"] else "one launch sequence per view",
                       "per_view_densification_stats_in_step": True,
                       "parallelism": f"frame-parallel dp{world}, scene replica per GPU"
                                      + (", per step ONE SUM all-reduce over [accumulated parameter grads | densification-statistics "
                                         "increments] + ONE MAX all-reduce of max_radii2D" if world > 1 else "")},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "traffic_source": ("profiles/traffic.json (builder rocprofv3 PMC run of this workload and launch mode; "
                                            "a replay, not measured live)") if traffic is not None else None,
                         "frames_per_launch": vpl,
                         "algorithmic_bytes_per_launch": int(st_bytes[dom]) * vpl, "avg_ms": round(dom_avg, 4),
                         "note": "composite kernels are VALU-issue-bound, not HBM-bound (DESIGN.md); see roofline_valu"},
            "roofline_valu": roofline_valu,
            "frame_hbm": {"algorithmic_bytes_per_frame": frame_bytes, "lineage_radix_passes": n_pass,
                          "achieved_GBps": round(frame_bytes * value / world / 1e9, 1),
                          "frac_of_peak": round(frame_bytes * value / world / 1e9 / HBM_PEAK_GBS, 5),
                          "ms_per_frame": round(ms_per_step / max(len(views), 1), 4)},
            "frame_valu": valu,
            # the same step as the reference's own loop of per-view GaussianRasterizer calls (secondary; None when that IS the run)
            "per_view_loop": next(({"value": m["value"], "ms_per_step": m["ms_per_step"], "unit": "frames/s"}
                                   for m in multi_stream if m["streams"] == 1), None),
            "multi_stream": multi_stream or None,
            "stages": per_stage,
            "stages_note": f"per-stage table from {BREAKDOWN_STEPS} untimed steps with every stage bracketed by HIP "
                           f"events; only '{dom}' is bracketed inside the timed region. lineage_bytes = SURVEY 8d's "
                           "compulsory figure for the lineage's formulation of the stage; actual_bytes = compulsory "
                           "bytes of this implementation's stage (actual_GBps is quoted on it; P-sized inputs are "
                           "shared by the 5 views of a step and may be served by the 256-MiB Infinity Cache)",
        }
        if DIST_ON:
            red = info.get("reduce_path") or {}
            out["rccl_ranks"] = dist.get_world_size()
            out["dist_backend"] = dist.get_backend()
            out["dist_env"] = dist_env_report()
            out["reduce_path"] = red.get("sum_path")
            out["collectives_per_step"] = red.get("collectives")
            out["config"]["grad_allreduce_path"] = red
        if args.fwd_only:
            out["metric"] = "DEBUG fwd-only frames/s (not the BASELINE metric)"
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload)
            if args.workload == "S0":      # BASELINE config 1: the PyTorch-CPU autograd reference, timed beside the oracle
                out["cpu_baseline"]["pytorch_cpu_autograd"] = cpu_baseline_torch_dense("S0")
        print(json.dumps(out), flush=True)
    if DIST_ON:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
