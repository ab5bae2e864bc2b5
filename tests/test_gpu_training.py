"""End-to-end checks on the GPU: the gradients drive an optimiser the way train_gaussians.py
uses them, and bench.py's multi-rank path runs (2 ranks on one GPU, gloo)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from splatloc_amd.synthetic import make_scene
from tests.helpers import hip_settings

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _render(rast, p, sc_scales, C):
    # activations as GaussianModel's getters apply them (gaussian_model.py:78-113)
    return rast(means3D=p["xyz"], means2D=torch.zeros_like(p["xyz"], requires_grad=True), shs=None,
                colors_precomp=torch.sigmoid(p["feat"]), opacities=torch.sigmoid(p["opacity"]),
                scales=torch.exp(p["log_scale"]), rotations=torch.nn.functional.normalize(p["rot"]),
                cov3D_precomp=None)


@pytest.mark.parametrize("C", [4, 35])
def test_adam_fits_a_target_render(C):
    """map()-shaped optimisation (train_gaussians.py:187-267): L1 on colour + depth, Adam on
    the same parameter groups; the loss must drop substantially within 60 steps."""
    from splatloc_amd import GaussianRasterizer
    dev = torch.device("cuda:0")
    sc = make_scene(3000, 160, 128, C, 5, scale_median=0.04)
    rast = GaussianRasterizer(raster_settings=hip_settings(sc, dev))
    g = torch.Generator().manual_seed(1)
    inv_sig = lambda x: torch.log(x / (1 - x))  # noqa: E731
    target = {"xyz": sc.means3D, "feat": inv_sig(sc.features.clamp(0.02, 0.98)),
              "opacity": inv_sig(sc.opacities.clamp(0.02, 0.98)), "log_scale": torch.log(sc.scales),
              "rot": sc.rotations}
    with torch.no_grad():
        tgt = {k: v.to(dev) for k, v in target.items()}
        t_color, t_depth, _, _ = _render(rast, tgt, None, C)
    p = {"xyz": (sc.means3D + 0.01 * torch.randn(sc.means3D.shape, generator=g)),
         "feat": target["feat"] + 0.8 * torch.randn(target["feat"].shape, generator=g),
         "opacity": target["opacity"] + 0.5 * torch.randn(target["opacity"].shape, generator=g),
         "log_scale": target["log_scale"] + 0.15 * torch.randn(target["log_scale"].shape, generator=g),
         "rot": target["rot"] + 0.1 * torch.randn(target["rot"].shape, generator=g)}
    p = {k: v.to(dev).requires_grad_(True) for k, v in p.items()}
    opt = torch.optim.Adam([{"params": [p["xyz"]], "lr": 2e-4}, {"params": [p["feat"]], "lr": 5e-2},
                            {"params": [p["opacity"]], "lr": 3e-2}, {"params": [p["log_scale"]], "lr": 5e-3},
                            {"params": [p["rot"]], "lr": 2e-3}], eps=1e-15)
    losses = []
    for _ in range(60):
        color, depth, alpha, radii = _render(rast, p, None, C)
        loss = (color - t_color).abs().mean() + (depth - t_depth).abs().mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        assert all(torch.isfinite(v.grad).all() for v in p.values())
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < 0.45 * losses[0], (losses[0], losses[-1])
    assert radii.dtype == torch.int32 and int((radii > 0).sum()) > 1000


def test_five_views_alive_before_one_backward():
    """train_gaussians.py:195-229 keeps window_size = 5 forwards alive, sums the losses, and runs
    ONE backward: saved state of different frames must not alias."""
    from splatloc_amd import GaussianRasterizer
    from splatloc_amd.camera import PinholeCamera
    dev = torch.device("cuda:0")
    sc = make_scene(4000, 192, 128, 4, 9, scale_median=0.04).to(dev)
    leaves = [t.clone().requires_grad_(True) for t in (sc.means3D, sc.features, sc.opacities, sc.scales, sc.rotations)]

    def view(k):
        ang = 0.05 * (k - 2)
        R = torch.tensor([[torch.cos(torch.tensor(ang)), 0, torch.sin(torch.tensor(ang))], [0, 1, 0],
                          [-torch.sin(torch.tensor(ang)), 0, torch.cos(torch.tensor(ang))]])
        cam = PinholeCamera(192, 128, 96.0, 96.0, 95.5, 63.5, R, torch.tensor([0.02 * k, 0.0, 0.0])).to(dev)
        sc.camera = cam
        return GaussianRasterizer(raster_settings=hip_settings(sc, dev))

    def render(k, ls):
        m3, col, op, sca, rot = ls
        return view(k)(means3D=m3, means2D=torch.zeros_like(m3, requires_grad=True), shs=None, colors_precomp=col,
                       opacities=op, scales=sca, rotations=rot, cov3D_precomp=None)

    g = torch.Generator().manual_seed(0)
    ws = [torch.rand(4, 128, 192, generator=g).to(dev) for _ in range(5)]
    total = 0
    for k in range(5):                      # five forwards alive
        color, depth, alpha, _ = render(k, leaves)
        total = total + (color * ws[k]).sum() + depth.sum() * 1e-3
    total.backward()                        # one backward
    joint = [l.grad.clone() for l in leaves]
    for l in leaves:
        l.grad = None
    for k in range(5):                      # reference: one view at a time
        color, depth, alpha, _ = render(k, leaves)
        ((color * ws[k]).sum() + depth.sum() * 1e-3).backward()
    for a, b in zip(joint, [l.grad for l in leaves]):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-3 * scale + 1e-12


def test_parameter_grads_of_a_frame_form_one_span():
    """The backward hands autograd all parameter gradients as pieces of one allocation, and
    autograd keeps them (no clone), so frame_parallel.allreduce_grads reduces ONE buffer in
    place; the viewspace gradient is separate."""
    from splatloc_amd import GaussianRasterizer
    from splatloc_amd.frame_parallel import _shared_spans
    dev = torch.device("cuda:0")
    P, C = 3001, 35                          # odd P: pieces need alignment padding
    sc = make_scene(P, 160, 96, C, 4, scale_median=0.04).to(dev)
    leaves = [t.clone().requires_grad_(True) for t in (sc.means3D, sc.features, sc.opacities, sc.scales, sc.rotations)]
    m2 = torch.zeros_like(leaves[0], requires_grad=True)
    rast = GaussianRasterizer(raster_settings=hip_settings(sc, dev))
    for rep in range(2):                     # second backward accumulates in place into the same span
        color, depth, alpha, _ = rast(means3D=leaves[0], means2D=m2, shs=None, colors_precomp=leaves[1],
                                      opacities=leaves[2], scales=leaves[3], rotations=leaves[4], cov3D_precomp=None)
        (color.sum() + depth.sum()).backward()
        spans, rest = _shared_spans([l.grad for l in leaves])
        assert len(spans) == 1 and not rest
        n = sum((l.numel() + 3) & ~3 for l in leaves)
        assert n - 3 <= spans[0].numel() <= n
        assert m2.grad.untyped_storage().data_ptr() != leaves[0].grad.untyped_storage().data_ptr()
        for l in leaves:
            assert l.grad.data_ptr() % 16 == 0 and torch.isfinite(l.grad).all()


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_on_one_gpu_gloo(scaling):
    """bench.py --gpus 2 through torch.distributed.run (the driver's launch line), gloo backend,
    both ranks on cuda:0: exercises rank/world plumbing, the per-rank views, the gradient all-reduce
    (in-place span path), the densification-statistics sync and the max-over-ranks timing, in the weak
    (own window per rank) and the strong (one window of 3 views dealt to 2 ranks) mode.
    (RCCL itself needs 2 GPUs; this box has one.)"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SPLATLOC_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    views = 2 if scaling == "weak" else 3
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--workload", "S0", "--no-cpu-baseline", "--views", str(views), "--scaling", scaling]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]     # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == scaling
    assert out["config"]["frames_per_step"] == (4 if scaling == "weak" else 3)
    assert out["config"]["views_on_rank0_per_step"] == 2          # weak: its own 2; strong: views 0 and 2 of 3
    path = out["config"]["grad_allreduce_path"]
    # the gradients AND the statistics increments were reduced where the backward left them, as ONE SUM; + ONE MAX
    assert path["sum_path"] == "in-place span" and path["collectives"] == 2, path
    assert out["rccl_ranks"] == 2 and out["collectives_per_step"] == 2 and out["reduce_path"] == "in-place span"
    assert out["repeats"]["regions"] == 5 and len(out["repeats"]["ms_per_step_all"]) == 5
    assert out["value"] > 0 and out["steps"] == 3 and "roofline" in out and "cpu_baseline" not in out
    assert len(set(out["config"]["tile_instances_R_per_view"])) == 2   # different cameras -> different lists


def test_bench_gpus_flag_alone_launches_the_ranks():
    """`python bench.py --gpus 2` — the form of the driver's command, NO launcher around it — must start 2 ranks itself
    (round-3 verdict: `--gpus` was parsed and never read; the run printed n_gpus: 1).  gloo, both ranks on cuda:0.  One view in
    strong mode: rank 1 has NO view and contributes zeros in the layout of rank 0's backward (world size > window size)."""
    env = dict(os.environ, SPLATLOC_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "S0",
           "--no-cpu-baseline", "--views", "1", "--scaling", "strong", "--repeats", "2"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["config"]["frames_per_step"] == 1
    assert out["collectives_per_step"] == 2 and out["reduce_path"] == "in-place span"
    assert out["value"] > 0


def test_replicas_stay_bit_identical_when_a_rank_has_no_view():
    """Round-3 advisor finding: with more ranks than views (8 GPUs, 5-view window) a rank without work raised before the
    collectives and the others hung.  2 ranks, windows of ONE view: rank 1 renders nothing, contributes zeros, takes the same
    optimizer / densify / reset steps, and ends bit-identical to rank 0."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SPLATLOC_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tools", "replica_check.py"), "--steps", "4",
           "--window", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1500:], r.stderr[-2500:])
    out = json.loads(lines[0])
    assert out["world"] == 2 and out["identical"] and not out["mismatched"] and out["window"] == 1
    assert out["collectives_per_step"] == 2


def test_replicas_stay_bit_identical_through_map_steps_densify_and_reset():
    """Frame-parallel readiness without a second GPU: 2 ranks (gloo, both on cuda:0), every rank a
    replica of the scene, 4 map steps of splatloc_amd.training.map_step with the 5 views of each window dealt to the
    ranks, one densify_and_prune (counter-based split noise) and one opacity reset in between.  Parameters, Adam
    moments and step counters, statistics and row counts must be BIT-identical on both ranks afterwards
    (tools/replica_check.py compares sha256 digests): the replicas exchange reduced gradients / statistics only."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SPLATLOC_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tools", "replica_check.py"), "--steps", "4"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1500:], r.stderr[-2500:])
    out = json.loads(lines[0])
    assert out["world"] == 2 and out["identical"] and not out["mismatched"]
    assert out["collectives_per_step"] == 2                             # ONE SUM [grads | statistics] + ONE MAX [radii | seen]
    assert out["rows_per_step"][1] != out["rows_per_step"][0]           # the densification really changed the model
    assert out["tensors_compared"] >= 25


@pytest.mark.parametrize("stage,extra", [("activations", []), ("loss", []), ("map_step", ["--workload", "S0"]),
                                         ("refine_step", ["--workload", "S0"])])
def test_bench_secondary_stages_run(stage, extra):
    """bench.py --stage ...: the §8f stage figures and the map()-shaped step stay runnable and
    print ONE JSON line with the contract's keys."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--stage", stage, "--steps", "3", "--warmup", "1"] + extra
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["value"] > 0 and out["unit"].endswith("/s") and "NOT the BASELINE metric" in out["metric"]
    if stage not in ("map_step", "refine_step"):
        assert 0 < out["roofline"]["frac"] < 1


def test_map_shaped_step_matches_reference_recording(golden_dir):
    """one SplatLoc.map step — 5 views x (render -> get_loss_mapping + get_loss_marker), the
    isotropic regulariser, ONE backward — recorded from the reference's own render() / GaussianModel / Camera /
    loss functions (tests/golden/make_golden_map_step.py; the oracle stands in for the un-vendored rasterizer),
    against the device path: fused front end, HIP rasterizer, fused losses."""
    import types
    from splatloc_amd.camera import PinholeCamera
    from splatloc_amd.fused import render
    from splatloc_amd.losses import isotropic_loss, mapping_loss
    from tests.helpers import assert_grad_close
    d = np.load(os.path.join(golden_dir, "map_step.npz"))
    dev = torch.device("cuda:0")
    fx, fy, cx, cy, W, H = (float(v) for v in d["intr"][:6])
    W, H = int(W), int(H)
    names = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
             "marker": "_marker", "kp_score": "_kp_score", "scaling": "_scaling", "rotation": "_rotation"}
    pc = types.SimpleNamespace(active_sh_degree=0, max_sh_degree=0)
    for k, a in names.items():
        setattr(pc, a, torch.from_numpy(d["raw_" + k]).to(dev).requires_grad_(True))
    assert pc._features_rest.shape[1:] == (0, 3)
    pipe = types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False)
    cfg = {"Training": {"primitive_reg": True, "rgb_boundary_threshold": 0.01}}
    bg = torch.zeros(3, device=dev)
    loss, pkgs, cams = 0, [], []
    for k in range(5):
        T = torch.from_numpy(d[f"view{k}_T"])
        cam = PinholeCamera(W, H, fx, fy, cx, cy, T[:3, :3], T[:3, 3]).to(dev)
        cam.original_image = torch.from_numpy(d[f"view{k}_color"]).to(dev)
        cam.depth = d[f"view{k}_depth"]                                   # numpy, as the reference's Camera holds it
        cam.kp_score = torch.from_numpy(d[f"view{k}_kp"]).to(dev)
        ea, eb = d[f"view{k}_exposure"]
        cam.exposure_a = torch.tensor([float(ea)], device=dev, requires_grad=True)
        cam.exposure_b = torch.tensor([float(eb)], device=dev, requires_grad=True)
        pkg = render(cam, pc, pipe, bg)
        loss = loss + mapping_loss(cfg, pkg["render"], pkg["depth"], pkg["kp_prob"], cam)
        pkgs.append(pkg)
        cams.append(cam)
    loss = loss + 0.01 * isotropic_loss(torch.exp(pc._scaling), pc._marker)
    loss.backward()
    np.testing.assert_allclose(float(loss.detach()), float(d["loss"]), rtol=2e-5)
    assert pc._marker.grad is None and not bool(d["has_grad_marker"])     # train_gaussians.py never reaches the marker
    for k, a in names.items():
        if not bool(d["has_grad_" + k]):
            continue
        got = getattr(pc, a).grad
        assert got is not None, k
        if d["grad_" + k].size == 0:
            assert got.numel() == 0
            continue
        assert_grad_close(k, got.cpu().numpy(), d["grad_" + k])
    for k in range(5):
        assert np.array_equal(pkgs[k]["radii"].cpu().numpy(), d[f"view{k}_radii"])
        assert_grad_close(f"viewspace {k}", pkgs[k]["viewspace_points"].grad.cpu().numpy(), d[f"view{k}_viewspace_grad"])
        ge = np.array([float(cams[k].exposure_a.grad), float(cams[k].exposure_b.grad)])
        np.testing.assert_allclose(ge, d[f"view{k}_exposure_grad"], rtol=2e-3, atol=1e-6)


def test_render_window_streams_match_serial_loop(golden_dir):
    """render_window on 1 / 2 / 3 HIP streams with the view-independent activations shared by the window == the serial loop of
    render() calls with one activate_pack per view: same images, and the same parameter
    gradients after ONE backward of the summed per-view losses (autograd replays each view's backward on its
    forward's stream).  Inputs: the reference-recorded map() step (tests/golden/map_step.npz)."""
    import types
    from splatloc_amd.camera import PinholeCamera
    from splatloc_amd.fused import render_window
    from splatloc_amd.losses import mapping_loss
    from tests.helpers import assert_grad_close
    d = np.load(os.path.join(golden_dir, "map_step.npz"))
    dev = torch.device("cuda:0")
    fx, fy, cx, cy, W, H = (float(v) for v in d["intr"][:6])
    W, H = int(W), int(H)
    names = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
             "marker": "_marker", "kp_score": "_kp_score", "scaling": "_scaling", "rotation": "_rotation"}
    pipe = types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False)
    cfg = {"Training": {"primitive_reg": True, "rgb_boundary_threshold": 0.01}}
    bg = torch.zeros(3, device=dev)
    results = {}
    for streams in (0, 1, 2, 3, "batched"):   # 0: the serial loop with one activate_pack per view (the baseline); "batched": one launch sequence per window
        pc = types.SimpleNamespace(active_sh_degree=0, max_sh_degree=0)
        for k, a in names.items():
            setattr(pc, a, torch.from_numpy(d["raw_" + k]).to(dev).requires_grad_(True))
        cams = []
        for k in range(5):
            T = torch.from_numpy(d[f"view{k}_T"])
            cam = PinholeCamera(W, H, fx, fy, cx, cy, T[:3, :3], T[:3, 3]).to(dev)
            cam.original_image = torch.from_numpy(d[f"view{k}_color"]).to(dev)
            cam.depth = d[f"view{k}_depth"]
            cam.kp_score = torch.from_numpy(d[f"view{k}_kp"]).to(dev)
            cam.exposure_a = torch.zeros(1, device=dev, requires_grad=True)
            cam.exposure_b = torch.zeros(1, device=dev, requires_grad=True)
            cams.append(cam)
        for rep in range(3):      # several windows back to back: stream reuse across windows
            for a in names.values():
                getattr(pc, a).grad = None
            batched = streams == "batched"
            pkgs, losses = render_window(cams, pc, pipe, bg, streams=1 if batched else max(streams, 1),
                                         share_activations=batched or streams > 0, batched=batched,
                                         per_view=lambda k, vp, pkg: mapping_loss(cfg, pkg["render"], pkg["depth"], pkg["kp_prob"], vp))
            sum(losses).backward()
        torch.cuda.synchronize()
        results[streams] = (torch.stack([p["render"] for p in pkgs]).cpu(), [p["radii"].cpu() for p in pkgs],
                            {a: getattr(pc, a).grad.cpu().numpy() for a in names.values() if getattr(pc, a).grad is not None
                             and getattr(pc, a).grad.numel()})
    for streams in (1, 2, 3, "batched"):
        assert torch.equal(results[streams][0], results[0][0])               # forward: bit-identical
        assert all(torch.equal(a, b) for a, b in zip(results[streams][1], results[0][1]))
        assert set(results[streams][2]) == set(results[0][2])
        for a, g in results[0][2].items():
            assert_grad_close(a, results[streams][2][a], g)                   # float atomics reorder sums
