"""The whole reconstruction schedule as ONE run (VERDICT r3 "missing" #3; train_gaussians.py:310-355 `do_recon`):
K key-frames x (extend_from_pcd_seq + 10 map iterations with densify_and_prune / reset_opacity_nonvisible on their
schedules), then the colour refinement, save_ply / load_ply and the forward-only eval loop with device PSNR / SSIM —
chained for thousands of iterations with P growing from zero.  Every piece is pinned by a reference-recorded fixture of a
few iterations elsewhere; this is the test that runs them in sequence on synthetic RGB-D key-frames rendered from a
ground-truth Gaussian set (splatloc_amd.scene.synthetic_keyframes).  Needs an MI355X."""
import copy
import json
import os
import socket
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _config(reset=137, every=40, offset=15):
    from splatloc_amd.scene import DEFAULT_CONFIG
    cfg = copy.deepcopy(DEFAULT_CONFIG)
    # the reference's schedule (densify every 150 / offset 50, reset every 2001) compressed so that a 20-key-frame run
    # (200 map iterations) sees several densifications AND an opacity reset
    cfg["Training"].update(gaussian_update_every=every, gaussian_update_offset=offset, gaussian_reset=reset)
    return cfg


def _psnr(model, frames, pipe, bg):
    from splatloc_amd.evaluation import eval_rendering
    return eval_rendering(frames, model, [f.original_image for f in frames], pipe, bg, window=5)


def test_whole_schedule_runs_and_reconstructs(tmp_path):
    from splatloc_amd.ply import load_ply, save_ply
    from splatloc_amd.scene import SceneModel, do_recon, state_digest, synthetic_keyframes
    dev = torch.device(DEV)
    W, H, K = 320, 240, 20
    frames, truth = synthetic_keyframes(K, W, H, P_truth=30_000, seed=1, device=dev)
    assert all(float((f.depth > 0).float().mean()) > 0.9 for f in frames)           # the camera looks at the room
    cfg = _config()
    pipe = types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False)
    bg = torch.zeros(3, device=dev)
    model = SceneModel(cfg, dev)
    trace = []

    def on_event(kind, k, gm):
        if kind == "keyframe" and k in (0, 9, 19):
            trace.append(("kf", k, gm.num_points, _psnr(gm, frames[:k + 1], pipe, bg)["mean_psnr"]))
        if kind == "refine" and k % 1000 == 0:
            trace.append(("refine", k, gm.num_points, _psnr(gm, frames, pipe, bg)["mean_psnr"]))

    torch.cuda.reset_peak_memory_stats(dev)
    stats = do_recon(model, frames, pipe, bg, cfg, refine_iterations=2000, seed=3, on_event=on_event)
    # ---- the schedule really ran: growth per key-frame, densifications, a reset, both loops ----
    assert stats["map_iterations"] == K * 10 and stats["refine_iterations"] == 2000
    rows = stats["rows_after_keyframe"]
    assert rows[0] > 500 and rows[-1] > 3 * rows[0]
    assert len(stats["densify_rows"]) >= 3 and any(b != a for _, a, b in stats["densify_rows"])
    assert stats["resets"] == 1
    assert stats["peak_memory_bytes"] < 8 << 30
    for k in ("_xyz", "_features_dc", "_opacity", "_scaling", "_rotation", "_kp_score", "_marker"):
        assert torch.isfinite(getattr(model, k)).all(), k
    for grp in model.optimizer.param_groups:
        st = model.optimizer.state.get(grp["params"][0])
        if st and grp["name"] != "f_rest":
            assert torch.isfinite(st["exp_avg"]).all() and torch.isfinite(st["exp_avg_sq"]).all(), grp["name"]
            assert st["exp_avg"].shape == grp["params"][0].shape
    n = model.num_points
    assert model.xyz_gradient_accum.shape == (n, 1) and model.denom.shape == (n, 1) and model.max_radii2D.shape == (n,)
    # ---- and it reconstructs: PSNR over all key-frames rises past a stated bar ----
    final = _psnr(model, frames, pipe, bg)
    kf = [t for t in trace if t[0] == "kf"]
    rf = [t for t in trace if t[0] == "refine"]
    assert len(kf) == 3 and len(rf) == 2
    assert final["mean_psnr"] >= 24.0, (final["mean_psnr"], trace)                  # synthetic room, 320 x 240
    assert final["mean_psnr"] >= rf[0][3] - 0.3 and rf[0][3] > 18.0, trace           # the refinement does not undo the map
    assert final["mean_ssim"] > 0.7
    assert min(final["psnr"]) > 18.0                                                  # every key-frame, not the mean only
    # ---- the artefact: point_cloud.ply round trip reproduces the renders ----
    path = os.path.join(str(tmp_path), "point_cloud.ply")
    save_ply(model, path)
    clone = SceneModel(cfg, dev)
    load_ply(clone, path, device=dev)
    assert clone._xyz.shape == model._xyz.shape
    again = _psnr(clone, frames, pipe, bg)
    assert abs(again["mean_psnr"] - final["mean_psnr"]) <= 1e-3 and abs(again["mean_ssim"] - final["mean_ssim"]) <= 1e-5
    assert len(state_digest(model)) == 64


def test_per_view_drop_in_path_and_window_path_end_close():
    """The literal drop-in (the reference's loop of per-view render() calls, what an unmodified train_gaussians.py issues)
    and the window-batched path run the same schedule; forward results are bit-identical per view and gradients equal to
    float-atomic rounding, but Adam with eps = 1e-15 amplifies rounding on near-zero gradients, so the two runs are compared
    by what they reconstruct: same row counts at every key-frame until the first densification, PSNR within 0.5 dB."""
    from splatloc_amd.scene import SceneModel, do_recon, synthetic_keyframes
    dev = torch.device(DEV)
    frames, _ = synthetic_keyframes(8, 256, 192, P_truth=15_000, seed=2, device=dev)
    cfg = _config(reset=61, every=30, offset=10)
    pipe = types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False)
    bg = torch.zeros(3, device=dev)
    out = {}
    for batched in (True, False):
        model = SceneModel(cfg, dev)
        stats = do_recon(model, frames, pipe, bg, cfg, refine_iterations=300, seed=5, batched=batched)
        out[batched] = (stats, _psnr(model, frames, pipe, bg))
    (sa, pa), (sb, pb) = out[True], out[False]
    # the two runs really took different paths (round 4's first version patched a function map_step no longer called:
    # both runs went through the graph-free window path and the comparison was vacuous)
    assert sa["render_paths"] == ["direct-window"] and sb["render_paths"] == ["per-view"], (sa["render_paths"], sb["render_paths"])
    assert sa["rows_after_keyframe"][:1] == sb["rows_after_keyframe"][:1]
    assert abs(sa["rows_final"] - sb["rows_final"]) <= 0.05 * sa["rows_final"]
    assert abs(pa["mean_psnr"] - pb["mean_psnr"]) <= 0.5, (pa["mean_psnr"], pb["mean_psnr"])
    assert pa["mean_psnr"] > 20.0


def test_replicas_run_the_whole_schedule_bit_identically():
    """2 ranks (gloo, both on cuda:0), every rank a replica: the whole schedule — key-frame insertion with the keyed
    down-sampling draw, map steps with the views dealt to the ranks, densifications, a reset, the refinement — must leave
    bit-identical replicas (sha256 of every parameter / moment / statistic)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SPLATLOC_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", "scene_replica_check.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1500:], r.stderr[-2500:])
    out = json.loads(lines[0])
    assert out["world"] == 2 and out["identical"], out
    assert out["rows_final"] > out["rows_after_keyframe"][0] and out["densifications"] >= 1


def _bench_scene_two_ranks(extra):
    env = dict(os.environ, SPLATLOC_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--stage", "scene", "--gpus", "2", "--keyframes", "6", "--refine", "40",
           "--truth", "20000"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1500:], r.stderr[-2500:])
    return json.loads(lines[0])


def test_bench_scene_replicas_one_scene_per_rank():
    """BASELINE config 5 as /root/reference/replica.sh:1-6 has it — independent scenes — on 2 ranks (gloo, both on cuda:0):
    `bench.py --stage scene --gpus 2 --replicas` runs one do_recon per rank with NO collective on the data path (a process
    group exists, for the launch and the report only); the scenes differ (seed = rank); value = 2 scenes / the slower rank."""
    out = _bench_scene_two_ranks(["--replicas"])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["multi_gpu_mode"] == "replicas"
    assert out["config"]["scenes"] == 2 and out["config"]["collectives_on_the_data_path"] == 0
    pr = out["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1] and all(r["rows_final"] > 0 and r["mean_psnr"] > 15.0 for r in pr)
    assert pr[0]["rows_final"] != pr[1]["rows_final"]          # two different rooms
    t_max = max(r["recon_s"] for r in pr)
    assert abs(out["value"] - 2 * 3600.0 / t_max) <= 1e-3 * out["value"] + 1e-3


@pytest.mark.parametrize("reduce", ["ring", "rs_ag"])
def test_bench_scene_frame_parallel(reduce):
    """The same schedule as ONE scene reconstructed by 2 ranks (views of every map window dealt to the ranks; gloo: rs_ag is
    emulated by the all-reduce, the padding / packing path is real)."""
    out = _bench_scene_two_ranks(["--reduce", reduce])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["multi_gpu_mode"] == "frame-parallel"
    assert out["config"]["scenes"] == 1 and out["config"]["reduce"] == reduce
    pr = out["per_rank"]
    assert pr[0]["rows_final"] == pr[1]["rows_final"] and abs(pr[0]["mean_psnr"] - pr[1]["mean_psnr"]) < 1e-6   # replicas
    assert out["eval"]["mean_psnr"] > 15.0
