#!/usr/bin/env python3
"""Per-tile list lengths and the refinement iteration's kernel table on a RECONSTRUCTED synthetic room (splatloc_amd.scene):
what the tile lists of a map look like (a room is not the uniform cloud of the S* workloads), and which front end serves them.
usage: python tools/scene_lists.py [keyframes=60] [truth=200000] [refine=300] [W=640] [H=480]      (front end: SPLATRASTER_FRONT_END)"""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    from splatloc_amd import _native, introspect
    from splatloc_amd.scene import DEFAULT_CONFIG, SceneModel, do_recon, synthetic_keyframes
    from splatloc_amd.training import color_refinement_step
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    truth = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
    refine = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    dev = torch.device("cuda:0")
    W = int(sys.argv[4]) if len(sys.argv) > 4 else 640
    H = int(sys.argv[5]) if len(sys.argv) > 5 else 480
    frames, _ = synthetic_keyframes(K, W, H, P_truth=truth, seed=0, device=dev)
    pipe = types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False)
    bg = torch.zeros(3, device=dev)
    model = SceneModel(DEFAULT_CONFIG, dev)
    do_recon(model, frames, pipe, bg, DEFAULT_CONFIG, refine_iterations=refine, seed=0)
    P = int(model._xyz.shape[0])
    out = {"keyframes": K, "rows": P, "front_end": os.environ.get("SPLATRASTER_FRONT_END", "-1"), "views": []}
    from splatloc_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from splatloc_amd.fused import activate_pack
    with torch.no_grad():
        sca, rot, opa, col = activate_pack(model._xyz, model._features_dc, model._features_rest, model._scaling, model._rotation,
                                           model._opacity, extra=model._kp_score)
        act = {"scales": sca, "rotations": rot, "opacities": opa, "colors": col}
    for f in frames[:: max(K // 6, 1)]:
        rs = GaussianRasterizationSettings(H, W, f.tanfovx, f.tanfovy, bg, 1.0, f.world_view_transform, f.full_proj_transform,
                                           0, f.camera_center, False, False)
        m3 = model._xyz.detach().requires_grad_(True)
        color, depth, alpha, radii = GaussianRasterizer(raster_settings=rs)(
            means3D=m3, means2D=torch.zeros_like(m3), shs=None, colors_precomp=act["colors"], opacities=act["opacities"],
            scales=act["scales"], rotations=act["rotations"], cov3D_precomp=None)
        fn = color.grad_fn
        sv = fn.saved_tensors
        st = introspect.forward_state((sv[12], sv[13], sv[14]), P, W, H, fn.num_rendered)
        lens = (st["ranges"][:, 1] - st["ranges"][:, 0]).long().cpu()
        out["views"].append({"uid": f.uid, "R": int(fn.num_rendered), "mean": round(float(lens.float().mean()), 1),
                             "max": int(lens.max()), "over_1024": int((lens > 1024).sum()), "over_2048": int((lens > 2048).sum()),
                             "over_4096": int((lens > 4096).sum()), "over_16384": int((lens > 16384).sum())})
    # kernel table of refinement iterations on this model
    from torch.profiler import ProfilerActivity, profile
    it = [0]

    def loop(n):
        for _ in range(n):
            it[0] += 1
            color_refinement_step(frames[it[0] % K], model, pipe, bg, 0.2, it[0])
    loop(30)
    torch.cuda.synchronize()
    import time
    regions = []
    for _ in range(int(os.environ.get("SCENE_LISTS_REGIONS", "1"))):      # (A/B runs: several regions, the median is reported)
        t0 = time.perf_counter()
        loop(300)
        torch.cuda.synchronize()
        regions.append(round((time.perf_counter() - t0) / 300 * 1e6, 1))
    out["refine_us_per_iteration"] = sorted(regions)[len(regions) // 2]
    out["refine_us_all_regions"] = regions
    M = 60
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        loop(M)
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages():
        dt = getattr(e, "device_time_total", None)
        if dt is None:
            dt = getattr(e, "cuda_time_total", 0)
        if dt and e.count:
            rows.append((e.key[:60], e.count / M, dt / M))
    rows.sort(key=lambda r: -r[2])
    out["kernels"] = [{"kernel": k, "launches": round(c, 2), "us": round(u, 1)} for k, c, u in rows[:16]]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
