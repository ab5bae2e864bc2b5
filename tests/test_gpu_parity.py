"""Parity of the HIP path (through the C ABI) against the CPU oracle — needs an MI355X.

Bar (BASELINE.json north_star): bit-exact radii / tile counts / point list / tile ranges /
n_contrib / final_T / alpha, <= 1e-4 abs on colour / depth buffers; gradients within 2e-3
relative + 1e-4 of the tensor's scale (float atomics reorder the sums).
"""
import json
import os

import numpy as np
import pytest
import torch

from splatloc_amd.synthetic import make_scene
from tests.helpers import HipRun, assert_grad_close, oracle_backward, oracle_forward

pytestmark = pytest.mark.gpu

IMG_TOL = 1e-4


def _check_forward(run: HipRun, f: dict, sc):
    P = sc.means3D.shape[0]
    W, H = sc.camera.image_width, sc.camera.image_height
    st = run.state
    # ---- bit-exact integer / index outputs ----
    assert np.array_equal(run.np(run.radii), f["radii"]), "radii"
    assert np.array_equal(run.np(st["tiles_touched"]).astype(np.uint32), f["tiles_touched"]), "tiles_touched"
    assert run.num_rendered == f["num_rendered"], "num_rendered"
    vis = f["radii"] > 0
    rec0 = run.np(st["rec0"])
    assert np.array_equal(rec0[vis, 0].view(np.uint32), f["xy"][vis, 0].view(np.uint32)), "pixel x bits"
    assert np.array_equal(rec0[vis, 1].view(np.uint32), f["xy"][vis, 1].view(np.uint32)), "pixel y bits"
    assert np.array_equal(rec0[vis, 2].view(np.uint32), f["view_depth"][vis].view(np.uint32)), "depth bits"
    assert np.array_equal(run.np(st["point_list"]).astype(np.uint32), f["point_list"]), "sorted point list"
    assert np.array_equal(run.np(st["ranges"]).astype(np.uint32), f["ranges"]), "tile ranges"
    tile_of = (f["keys"] >> np.uint64(32)).astype(np.uint32)
    assert np.array_equal(run.np(st["tile_list"]).astype(np.uint32), tile_of), "tile ids"
    # ---- float buffers ----
    np.testing.assert_allclose(run.np(st["rec1"])[vis], f["conic_opacity"][vis], rtol=1e-6, atol=0)
    assert np.abs(run.np(run.color) - f["color"]).max() <= IMG_TOL
    assert np.abs(run.np(run.alpha) - f["alpha"]).max() <= IMG_TOL
    dscale = max(1.0, float(np.abs(f["depth"]).max()))
    assert np.abs(run.np(run.depth) - f["depth"]).max() <= IMG_TOL * dscale
    # alpha, the transmittance chain and every threshold decision follow ONE arithmetic contract on
    # both sides (exp2_shared / orc_exp2): the integer n_contrib and final_T are bit-exact
    nc = run.np(st["n_contrib"]).astype(np.int64)
    assert np.array_equal(nc, f["n_contrib"].astype(np.int64)), \
        f"n_contrib differs on {(nc != f['n_contrib']).sum()} pixels"
    assert np.array_equal(run.np(st["final_T"]).view(np.uint32), f["final_T"].view(np.uint32)), "final_T bits"
    assert np.array_equal(run.np(run.alpha)[0].view(np.uint32), f["alpha"][0].view(np.uint32)), "alpha bits"
    assert P == rec0.shape[0] and (H, W) == nc.shape


def _check_backward(run: HipRun, b: dict, allow_frac=0.0, full_size=False):
    """Every gradient against the oracle.  `full_size` (the BASELINE configurations): the tightened tensor bar (rtol 1e-4 +
    5e-5 of the tensor's scale; round 3: 2e-3 + 1e-4 — float atomics reorder the sums from run to run: one element in 1.5 M was
    seen at 3e-5) AND the per-ROW bar (rtol 1e-4 + 1e-3 of the row's own maximum) — the backward walks back to front (round 4),
    so a Gaussian's gradient is accurate relative to its own magnitude.  The one-element opacity rows make that a purely
    relative bar on a sum of signed G dL/dalpha terms: the Gaussians whose sum cancels are counted (<= 1e-3 of them), not bounded."""
    kw = dict(allow_frac=allow_frac)
    if full_size:
        from tests.helpers import assert_grad_rows_close
        kw = dict(rtol=1e-4, atol_scale=5e-5)
        for name, got, ref in (("dL_dmeans3D", run.means3D.grad, b["dL_dmeans3D"]), ("dL_dmeans2D", run.means2D.grad, b["dL_dmeans2D"]),
                               ("dL_dscales", run.scales.grad, b["dL_dscales"]), ("dL_drotations", run.rotations.grad, b["dL_drotations"])):
            assert_grad_rows_close("rows " + name, run.np(got), ref, rtol=1e-4, row_atol=1e-3, allow_frac=1e-4, outlier_factor=30.0)
        assert_grad_rows_close("rows dL_dcolors", run.np(run.colors.grad), b["dL_dcolors"], rtol=1e-4, row_atol=1e-5)
        assert_grad_rows_close("rows dL_dopacities", run.np(run.opacities.grad), b["dL_dopacities"], rtol=1e-4, row_atol=1e-3,
                               allow_frac=1e-3, outlier_factor=float("inf"))
    assert_grad_close("dL_dmeans3D", run.np(run.means3D.grad), b["dL_dmeans3D"], **kw)
    assert_grad_close("dL_dmeans2D", run.np(run.means2D.grad), b["dL_dmeans2D"], **kw)
    assert_grad_close("dL_dopacities", run.np(run.opacities.grad), b["dL_dopacities"], **kw)
    if run.colors is not None:
        assert_grad_close("dL_dcolors", run.np(run.colors.grad), b["dL_dcolors"], **kw)
    if run.shs is not None:
        assert_grad_close("dL_dshs", run.np(run.shs.grad), b["dL_dshs"], **kw)
    if run.scales is not None:
        assert_grad_close("dL_dscales", run.np(run.scales.grad), b["dL_dscales"], **kw)
        assert_grad_close("dL_drotations", run.np(run.rotations.grad), b["dL_drotations"], **kw)
    if run.cov3D is not None:
        assert_grad_close("dL_dcov3D", run.np(run.cov3D.grad), b["dL_dcov3D"], **kw)


@pytest.mark.parametrize("cfg", [
    dict(P=10_000, W=640, H=480, C=3, seed=0, scale_median=0.02),    # BASELINE config 1 shape (S0)
    dict(P=4_000, W=640, H=480, C=4, seed=21, scale_median=0.03),    # reference channel layout
    dict(P=3_000, W=333, H=201, C=35, seed=22, scale_median=0.03),   # north-star channel count, ragged edges
    dict(P=2_000, W=200, H=120, C=1, seed=23, scale_median=0.04),
    dict(P=2_000, W=200, H=120, C=7, seed=24, scale_median=0.04),    # generic C: chunked passes 4+3
    dict(P=1_500, W=160, H=96, C=40, seed=25, scale_median=0.04),    # 32 + 8
    dict(P=20_000, W=256, H=256, C=3, seed=26, scale_median=0.05),   # deep lists: several LDS batches
    dict(P=30_000, W=1280, H=720, C=4, seed=27, scale_median=0.02),  # C = 4 on a LARGE frame: the butterfly variant of the backward (small frames take the panel variant)
    dict(P=6_000, W=320, H=240, C=8, seed=28, scale_median=0.03),    # C = 8: panel variant + matrix-pipe dot products
])
def test_forward_backward_parity(cfg):
    sc = make_scene(**cfg)
    f = oracle_forward(sc)
    b = oracle_backward(f, sc)
    run = HipRun(sc)
    _check_forward(run, f, sc)
    _check_backward(run, b)


def test_scale_modifier_and_no_aux_grads():
    sc = make_scene(3000, 320, 240, 4, 31, scale_median=0.03)
    f = oracle_forward(sc, scale_modifier=1.6)
    b = oracle_backward(f, sc, use_depth=False, use_alpha=False)
    run = HipRun(sc, scale_modifier=1.6, use_depth=False, use_alpha=False)
    _check_forward(run, f, sc)
    _check_backward(run, b)


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_path(deg):
    sc = make_scene(2500, 256, 192, 3, 40 + deg, scale_median=0.03)
    g = torch.Generator().manual_seed(77 + deg)
    shs = 0.5 * torch.randn(2500, 16, 3, generator=g)
    f = oracle_forward(sc, sh_degree=deg, colors_precomp=None, shs=shs.numpy())
    b = oracle_backward(f, sc)
    run = HipRun(sc, sh_degree=deg, shs=shs)
    _check_forward(run, f, sc)
    assert np.array_equal(run.np(run.state["clamped"])[f["radii"] > 0], f["clamped"][f["radii"] > 0])
    _check_backward(run, b)


def test_cov3d_precomp_path():
    sc = make_scene(2500, 256, 192, 3, 50, scale_median=0.03)
    g = torch.Generator().manual_seed(5)
    L = torch.randn(2500, 3, 3, generator=g) * 0.03
    S = L @ L.transpose(1, 2)
    cov = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).contiguous()
    f = oracle_forward(sc, scales=None, rotations=None, cov3D_precomp=cov.numpy())
    b = oracle_backward(f, sc)
    run = HipRun(sc, cov3D=cov)
    _check_forward(run, f, sc)
    _check_backward(run, b)


def test_all_culled_and_empty():
    """Edge cases: every Gaussian behind the camera (R = 0) and P = 0."""
    sc = make_scene(500, 128, 96, 3, 60)
    sc.means3D[:, 2] = -sc.means3D[:, 2]
    sc.bg = torch.tensor([0.2, 0.4, 0.6])
    run = HipRun(sc)
    assert run.num_rendered == 0 and int(run.radii.abs().sum()) == 0
    col = run.np(run.color)
    assert np.allclose(col[0], 0.2) and np.allclose(col[1], 0.4) and np.allclose(col[2], 0.6)
    assert np.all(run.np(run.alpha) == 0) and np.all(run.np(run.depth) == 0)
    assert float(run.means3D.grad.abs().sum()) == 0.0 and float(run.colors.grad.abs().sum()) == 0.0

    from splatloc_amd import GaussianRasterizer
    from tests.helpers import hip_settings
    dev = torch.device("cuda:0")
    rast = GaussianRasterizer(raster_settings=hip_settings(sc, dev))
    e = lambda *s: torch.zeros(*s, device=dev)  # noqa: E731
    color, depth, alpha, radii = rast(means3D=e(0, 3), means2D=e(0, 3), shs=None, colors_precomp=e(0, 3),
                                      opacities=e(0, 1), scales=e(0, 3), rotations=e(0, 4), cov3D_precomp=None)
    assert color.shape == (3, 96, 128) and radii.numel() == 0
    assert torch.allclose(color[1], torch.full_like(color[1], 0.4))


def test_reference_boundary_inputs(golden_dir):
    """The exact tensors the reference's unmodified render() passes at the boundary
    (tests/golden/boundary.npz: C = 4, 3-entry bg, strided campos) through HIP vs oracle."""
    from oracle import oracle
    from splatloc_amd import GaussianRasterizationSettings, GaussianRasterizer
    g = np.load(os.path.join(golden_dir, "boundary.npz"))
    meta = json.loads(str(g["meta"]))
    s = meta["settings"]
    dev = torch.device("cuda:0")
    t = lambda k: torch.from_numpy(g[k]).to(dev)  # noqa: E731
    # reproduce the non-contiguous campos the reference hands over (camera_utils.py:139)
    campos = torch.zeros(3, 4, device=dev)
    campos[:, 3] = t("rs_campos")
    rs = GaussianRasterizationSettings(s["image_height"], s["image_width"], s["tanfovx"], s["tanfovy"], t("rs_bg"),
                                       s["scale_modifier"], t("rs_viewmatrix"), t("rs_projmatrix"), s["sh_degree"],
                                       campos[:, 3], s["prefiltered"], s["debug"])
    assert not rs.campos.is_contiguous()
    leaf = lambda k: t(k).requires_grad_(True)  # noqa: E731
    inp = {k: leaf(k) for k in ("means3D", "means2D", "colors_precomp", "opacities", "scales", "rotations")}
    color, depth, alpha, radii = GaussianRasterizer(raster_settings=rs)(shs=None, cov3D_precomp=None, **inp)
    st = oracle.Settings(s["image_height"], s["image_width"], s["tanfovx"], s["tanfovy"])
    f = oracle.forward(st, g["rs_bg"], g["means3D"], g["opacities"], g["rs_viewmatrix"], g["rs_projmatrix"],
                       g["rs_campos"], colors_precomp=g["colors_precomp"], scales=g["scales"],
                       rotations=g["rotations"], omp=True)
    assert np.array_equal(radii.cpu().numpy(), f["radii"]) and radii.dtype == torch.int32
    assert color.shape == (4, 480, 640) and depth.shape == (1, 480, 640) and alpha.shape == (1, 480, 640)
    assert np.abs(color.detach().cpu().numpy() - f["color"]).max() <= IMG_TOL
    assert np.abs(alpha.detach().cpu().numpy() - f["alpha"]).max() <= IMG_TOL
    # the loss shape of train_gaussians.py: image[:3], kp_prob = image[-1], depth; opacity unused
    gl = np.load(os.path.join(golden_dir, "loss.npz"))
    gen = torch.Generator().manual_seed(3)
    dcol = (torch.rand(4, 480, 640, generator=gen) - 0.5) / (480 * 640)
    ddep = (torch.rand(1, 480, 640, generator=gen) - 0.5) / (480 * 640)
    del gl
    ((color * dcol.to(dev)).sum() + (depth * ddep.to(dev)).sum()).backward()
    b = oracle.backward(f, dcol.numpy(), ddep.numpy(), None, omp=True)
    assert_grad_close("means3D", inp["means3D"].grad.cpu().numpy(), b["dL_dmeans3D"])
    assert_grad_close("means2D", inp["means2D"].grad.cpu().numpy(), b["dL_dmeans2D"])
    assert_grad_close("colors", inp["colors_precomp"].grad.cpu().numpy(), b["dL_dcolors"])
    assert_grad_close("opacities", inp["opacities"].grad.cpu().numpy(), b["dL_dopacities"])
    assert_grad_close("scales", inp["scales"].grad.cpu().numpy(), b["dL_dscales"])
    assert_grad_close("rotations", inp["rotations"].grad.cpu().numpy(), b["dL_drotations"])


def test_mark_visible():
    from oracle import oracle
    from splatloc_amd import GaussianRasterizer
    from tests.helpers import hip_settings
    sc = make_scene(5000, 128, 96, 3, 61)
    sc.means3D[::3, 2] *= -1
    sc.means3D[1::7, 2] = 0.2
    dev = torch.device("cuda:0")
    vis = GaussianRasterizer(raster_settings=hip_settings(sc, dev)).markVisible(sc.means3D.to(dev))
    ref = oracle.mark_visible(sc.means3D.numpy(), sc.camera.world_view_transform.numpy())
    assert vis.dtype == torch.bool and np.array_equal(vis.cpu().numpy(), ref)


@pytest.mark.parametrize("n,bits", [(1, 32), (63, 8), (4096, 13), (4097, 32), (100_003, 32), (1_000_000, 16),
                                    (1_100_000, 13), (3_000_001, 16)])  # > 256 blocks: histogram/scan/scatter passes
def test_radix_sort_is_stable_and_exact(n, bits):
    """splatraster_sort_pairs_u32 vs numpy stable argsort (bit-exact, ties keep input order)."""
    import ctypes as C
    from splatloc_amd import _native
    lib = _native.load()
    rng = np.random.default_rng(n)
    hi = (1 << bits) - 1
    keys = rng.integers(0, min(hi, 5000 if n > 4096 else hi), size=n, endpoint=True, dtype=np.uint64).astype(np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    dev = torch.device("cuda:0")
    k = torch.from_numpy(keys.view(np.int32)).to(dev)
    v = torch.from_numpy(vals.view(np.int32)).to(dev)
    tmp = torch.empty(lib.splatraster_sort_tmp_bytes(n), dtype=torch.uint8, device=dev)
    _native.check(lib.splatraster_sort_pairs_u32(n, C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()), bits,
                                                 C.c_void_p(tmp.data_ptr()),
                                                 C.c_void_p(torch.cuda.current_stream().cuda_stream)), "sort")
    torch.cuda.synchronize()
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(k.cpu().numpy().view(np.uint32), keys[order])
    assert np.array_equal(v.cpu().numpy().view(np.uint32), vals[order])


@pytest.mark.parametrize("n", [1, 2, 3, 4, 257, 5000, 20_000])
def test_dist2_matches_oracle(n):
    """simple_knn distCUDA2 (gaussian_model.py:206): exact 3-NN mean squared distance."""
    from oracle import oracle
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(n)
    pts = rng.normal(size=(n, 3)).astype(np.float32)
    if n > 10:
        pts[5] = pts[4]          # duplicate point: distance 0 neighbour
    out = distCUDA2(torch.from_numpy(pts).cuda())
    ref = oracle.dist2(pts)
    assert out.shape == (n,) and out.dtype == torch.float32
    assert np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    if n >= 5000:
        from scipy.spatial import cKDTree
        d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
        np.testing.assert_allclose(out.cpu().numpy(), (d[:, 1:] ** 2).mean(1), rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize("n,kind", [(9, "uniform"), (1000, "uniform"), (20_000, "surface"), (40_000, "uniform"), (60_000, "plane"),
                                    (200_000, "surface"), (500_000, "clustered")])
def test_dist2_grid_equals_brute_force(n, kind):
    """The exact uniform-grid search (taken from 10 000 points; forced here at every size) against the tiled brute
    force: BIT-identical distances — same arithmetic, same 3 smallest values — on uniform clouds, a depth-map-like
    surface (what create_pcd_from_image produces), an exactly flat plane, clusters with duplicates; and against the
    CPU oracle where that is fast."""
    from oracle import oracle
    from simple_knn._C import distCUDA2
    from splatloc_amd import _native
    lib = _native.load()
    rng = np.random.default_rng(n)
    if kind == "uniform":
        pts = rng.random((n, 3)) * np.array([5.0, 3.0, 4.0])
    elif kind == "surface":
        xy = rng.random((n, 2)) * 6.0 - 3.0
        pts = np.concatenate([xy, (1.5 + 0.3 * np.sin(2 * xy[:, :1]) * np.cos(3 * xy[:, 1:]) + 0.002 * rng.normal(size=(n, 1)))], 1)
    elif kind == "plane":
        pts = np.concatenate([rng.random((n, 2)) * 4.0, np.full((n, 1), 2.5)], 1)
    else:
        centres = rng.normal(size=(40, 3)) * 3.0
        pts = centres[rng.integers(0, 40, n)] + rng.normal(size=(n, 3)) * rng.choice([0.01, 0.1, 0.5], size=(n, 1))
        pts[::1000] = pts[1::1000][: len(pts[::1000])]          # exact duplicates
    pts = torch.from_numpy(pts.astype(np.float32)).cuda()
    try:
        lib.splatknn_debug_set_grid_min(1 << 30)
        brute = distCUDA2(pts) if n <= 200_000 else None     # 500 k points: 2.5e11 evaluations — the oracle / brute leg stops at 200 k
        lib.splatknn_debug_set_grid_min(0)
        grid = distCUDA2(pts)
    finally:
        lib.splatknn_debug_set_grid_min(-1)
    default = distCUDA2(pts)
    assert torch.equal(default, grid if n >= 10_000 else (brute if brute is not None else grid))
    if brute is not None:
        assert torch.equal(grid.view(torch.int32), brute.view(torch.int32)), \
            f"{int((grid.view(torch.int32) != brute.view(torch.int32)).sum())} of {n} distances differ"
    if n <= 20_000:
        assert np.array_equal(grid.cpu().numpy().view(np.uint32), oracle.dist2(pts.cpu().numpy()).view(np.uint32))
    from scipy.spatial import cKDTree
    sub = rng.choice(n, size=min(n, 20_000), replace=False)
    p64 = pts.cpu().numpy().astype(np.float64)
    d, _ = cKDTree(p64).query(p64[sub], k=4)
    np.testing.assert_allclose(grid.cpu().numpy()[sub], (d[:, 1:] ** 2).mean(1), rtol=2e-4, atol=1e-9)


def test_dist2_degenerate_clouds_fall_back_to_the_brute_force():
    """Round-3 advisor finding: a cloud that collapses into a handful of grid cells (a dense cluster + a few far outliers that
    inflate the bounding box; mass duplicates) would make every query scan one huge cell serially.  The count pass records the
    largest cell; beyond max(4096, N / 64) points in one cell the tiled brute force — launched behind the grid query, returning
    at once otherwise — answers instead, decided on the device.  Same arithmetic: bit-identical to the forced brute force.
    Non-finite points are nobody's neighbour and do not inflate the box."""
    from simple_knn._C import distCUDA2
    from splatloc_amd import _native
    lib = _native.load()
    rng = np.random.default_rng(7)
    n = 30_000
    cluster = rng.normal(size=(n, 3)) * 1e-3 + np.array([0.3, -0.2, 2.0])
    cluster[-6:] = rng.normal(size=(6, 3)) * 5e3                 # six outliers: the box is 1e7 x the cluster
    dup = np.repeat(rng.normal(size=(30, 3)), 1000, axis=0)      # 30 sites x 1000 exact duplicates
    for name, pts in (("outliers", cluster), ("duplicates", dup)):
        t = torch.from_numpy(pts.astype(np.float32)).cuda()
        try:
            lib.splatknn_debug_set_grid_min(1 << 30)
            brute = distCUDA2(t)
            lib.splatknn_debug_set_grid_min(0)
            grid = distCUDA2(t)
        finally:
            lib.splatknn_debug_set_grid_min(-1)
        assert torch.equal(grid.view(torch.int32), brute.view(torch.int32)), name
        assert torch.equal(distCUDA2(t), brute), name
    assert float(distCUDA2(torch.from_numpy(dup.astype(np.float32)).cuda()).max()) == 0.0     # >= 3 duplicates everywhere
    # non-finite points: skipped as neighbours, their own result is 0, the others' results are those of the cloud without them
    pts = (rng.random((20_000, 3)) * 4.0).astype(np.float32)
    bad = pts.copy()
    bad[[5, 777, 19_999]] = [[np.inf, 0, 0], [np.nan, 1, 2], [0, -np.inf, 1]]
    good_idx = np.setdiff1d(np.arange(20_000), [5, 777, 19_999])
    ref = distCUDA2(torch.from_numpy(pts[good_idx]).cuda()).cpu().numpy()
    for force in (0, 1 << 30):
        try:
            lib.splatknn_debug_set_grid_min(force)
            out = distCUDA2(torch.from_numpy(bad).cuda()).cpu().numpy()
        finally:
            lib.splatknn_debug_set_grid_min(-1)
        assert np.array_equal(out[good_idx].view(np.uint32), ref.view(np.uint32)), force
        assert np.all(out[[5, 777, 19_999]] == 0.0)


def test_full_size_properties():
    """North-star shape (S2: 500k Gaussians, 1920x1080, C = 35): size-independent checks —
    sortedness of the instance list, range table partition, alpha = 1 - final_T,
    linearity of backward in dL/dout, preprocess bit-exact vs oracle."""
    from splatloc_amd.synthetic import make_workload
    sc = make_workload("S2")
    run = HipRun(sc, backward=True)
    st = run.state
    R = run.num_rendered
    tiles = st["tile_list"].long()
    assert R > 3_000_000 and bool((tiles[1:] >= tiles[:-1]).all())
    depth_bits = st["rec0"][:, 2].contiguous().view(torch.int32).long()
    pl = st["point_list"].long()
    same = tiles[1:] == tiles[:-1]
    d0, d1 = depth_bits[pl[:-1]], depth_bits[pl[1:]]
    assert bool(((d1 > d0) | ((d1 == d0) & (pl[1:] > pl[:-1])))[same].all()), "(depth, index) order inside tiles"
    rng = st["ranges"].long()
    assert int((rng[:, 1] - rng[:, 0]).sum()) == R
    nz = rng[:, 1] > rng[:, 0]
    assert bool((rng[nz][1:, 0] == rng[nz][:-1, 1]).all())
    assert float((run.alpha[0] - (1.0 - st["final_T"])).abs().max()) == 0.0
    assert bool(torch.isfinite(run.color).all()) and bool(torch.isfinite(run.means3D.grad).all())
    # preprocess + binning integers against the oracle (C oracle handles this size in seconds)
    from oracle import oracle
    cam = sc.camera
    f = oracle.forward(oracle.Settings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy),
                       sc.bg.numpy(), sc.means3D.numpy(), sc.opacities.numpy(), cam.world_view_transform.numpy(),
                       cam.full_proj_transform.numpy(), cam.camera_center.numpy(),
                       colors_precomp=sc.features[:, :1].contiguous().numpy(), scales=sc.scales.numpy(),
                       rotations=sc.rotations.numpy(), omp=True)
    assert np.array_equal(run.np(run.radii), f["radii"]) and R == f["num_rendered"]
    assert np.array_equal(run.np(st["point_list"]).astype(np.uint32), f["point_list"])
    assert np.array_equal(run.np(st["ranges"]).astype(np.uint32), f["ranges"])
    assert np.abs(run.np(run.alpha) - f["alpha"]).max() <= IMG_TOL
    # the full 35-channel image, n_contrib / final_T bit-exact and every gradient against the oracle
    # at full size (OpenMP build: seconds on the box) — no outlier allowance
    fb = oracle_forward(sc, omp=True)
    _check_forward(run, fb, sc)
    _check_backward(run, oracle_backward(fb, sc, omp=True), full_size=True)
    # linearity: backward(2 g) == 2 backward(g)
    g1 = run.means3D.grad.clone()
    c1 = run.colors.grad.clone()
    sc2 = sc
    sc2.dL_dcolor, sc2.dL_ddepth, sc2.dL_dalpha = 2 * sc.dL_dcolor, 2 * sc.dL_ddepth, 2 * sc.dL_dalpha
    run2 = HipRun(sc2, backward=True)
    assert_grad_close("linearity means3D", run2.np(run2.means3D.grad), 2 * run.np(g1), rtol=5e-3, atol_scale=2e-4)
    assert_grad_close("linearity colors", run2.np(run2.colors.grad), 2 * run.np(c1), rtol=5e-3, atol_scale=2e-4)


def _full_size_parity(name, backward=True):
    from splatloc_amd.synthetic import make_workload
    sc = make_workload(name)
    f = oracle_forward(sc, omp=True)
    run = HipRun(sc, backward=backward)
    _check_forward(run, f, sc)
    if backward:
        _check_backward(run, oracle_backward(f, sc, omp=True), full_size=True)
    return run, f


def test_full_size_S1():
    """BASELINE config 2 stand-in (S1: 300k Gaussians, 1200x680, RGB + depth + alpha, fwd+bwd)."""
    run, f = _full_size_parity("S1")
    assert run.num_rendered == f["num_rendered"] > 1_000_000


def test_full_size_S2_reference_layout():
    """What train_gaussians.py really renders: C = 4 ([rgb | kp_score]) at 640x480, at S2's 500k Gaussians."""
    _full_size_parity("S2-ref-layout")


def test_scenes12_forward_only_loop(golden_dir):
    """BASELINE config 4 stand-in: the eval_rendering loop shape (utils/eval_utils.py:22-72 — forward only,
    under no_grad, 640x480, 12-Scenes intrinsics fx = fy = 572, cx = 320, cy = 240,
    configs/scenes12/base_config.yaml:17-27) over the three reference-generated camera poses of
    tests/golden/camera.npz; every frame against the oracle."""
    from oracle import oracle
    from splatloc_amd import GaussianRasterizationSettings, GaussianRasterizer
    g = np.load(os.path.join(golden_dir, "camera.npz"))
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(412)
    P = 200_000
    # a box of Gaussians around the poses' viewing volume
    means = (torch.rand(P, 3, generator=gen) - 0.5) * torch.tensor([8.0, 6.0, 8.0])
    scales = torch.exp(np.log(0.012) + 0.5 * torch.randn(P, 3, generator=gen))
    q = torch.randn(P, 4, generator=gen)
    rots = q / q.norm(dim=1, keepdim=True)
    opac = torch.sigmoid(1.5 * torch.randn(P, 1, generator=gen))
    cols = torch.rand(P, 4, generator=gen)
    bg = torch.zeros(3)
    names = [k[:-len("_view")] for k in g.files if k.startswith("scenes12") and k.endswith("_view")]
    assert len(names) == 3, g.files
    total_R = 0
    with torch.no_grad():
        for nm in sorted(names):
            # world_view_transform / full_proj_transform / camera_center of the reference's Camera
            # (utils/camera_utils.py:129-139), intr = fx fy cx cy W H tanfovx tanfovy
            V, PM, cam = g[nm + "_view"], g[nm + "_fullproj"], g[nm + "_campos"]
            intr = g[nm + "_intr"]
            assert tuple(intr[:6]) == (572.0, 572.0, 320.0, 240.0, 640.0, 480.0)
            tfx, tfy = float(intr[6]), float(intr[7])
            V, PM, cam = (np.ascontiguousarray(a, dtype=np.float32) for a in (V, PM, cam))
            rs = GaussianRasterizationSettings(480, 640, tfx, tfy, bg.to(dev), 1.0, torch.from_numpy(V).to(dev),
                                               torch.from_numpy(PM).to(dev), 0, torch.from_numpy(cam).to(dev),
                                               False, False)
            color, depth, alpha, radii = GaussianRasterizer(raster_settings=rs)(
                means3D=means.to(dev), means2D=torch.zeros(P, 3, device=dev), shs=None, colors_precomp=cols.to(dev),
                opacities=opac.to(dev), scales=scales.to(dev), rotations=rots.to(dev), cov3D_precomp=None)
            f = oracle.forward(oracle.Settings(480, 640, tfx, tfy), bg.numpy(), means.numpy(), opac.numpy(), V, PM, cam,
                               colors_precomp=cols.numpy(), scales=scales.numpy(), rotations=rots.numpy(), omp=True)
            assert np.array_equal(radii.cpu().numpy(), f["radii"])
            assert np.abs(color.cpu().numpy() - f["color"]).max() <= IMG_TOL
            assert np.array_equal(alpha.cpu().numpy().view(np.uint32), f["alpha"].view(np.uint32))
            assert np.abs(depth.cpu().numpy() - f["depth"]).max() <= IMG_TOL * max(1.0, float(f["depth"].max()))
            total_R += f["num_rendered"]
    assert total_R > 100_000, total_R


def test_device_exp2_is_the_oracles_exp2_bit_for_bit():
    """exp2_shared (composite_common.h) vs orc_exp2 (splat_oracle.c) on 8 M arguments: the alpha arithmetic
    contract that makes n_contrib / final_T bit-exact."""
    import ctypes as C
    from oracle import oracle
    from splatloc_amd import _native
    lib = _native.load()
    rng = np.random.default_rng(5)
    x = np.concatenate([-(rng.random(6_000_000) * 24.0), -np.logspace(-38, 2.2, 1_000_000),
                        np.linspace(-130.0, 1.0, 1_000_001), np.array([0.0, -0.0, -0.5, -1.5, -2.5, -149.0, -151.0])
                        ]).astype(np.float32)
    dev = torch.device("cuda:0")
    xd = torch.from_numpy(x).to(dev)
    yd = torch.empty_like(xd)
    _native.check(lib.splatraster_debug_exp2(x.size, C.c_void_p(xd.data_ptr()), C.c_void_p(yd.data_ptr()),
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)), "debug_exp2")
    got = yd.cpu().numpy()
    ref = oracle.exp2(x)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), \
        f"{(got.view(np.uint32) != ref.view(np.uint32)).sum()} of {x.size} values differ"


@pytest.mark.parametrize("cfg", [
    dict(P=6_000, W=320, H=240, C=4, seed=81, scale_median=0.03),     # reference channel layout (VALU reductions)
    dict(P=4_000, W=333, H=201, C=35, seed=82, scale_median=0.03),    # matrix-pipe reductions + butterfly
])
def test_deterministic_sum_mode_is_bit_reproducible_and_strict(cfg):
    """Deterministic-sum debug mode (splatraster_debug_set_deterministic): the per-Gaussian gradient rows are summed
    with 64-bit integer atomics in 2^-40 fixed point, so two runs agree BIT FOR BIT (the float-atomic mode agrees
    only to rounding), and the gradients meet a 20x tighter bar against the oracle's double-precision sums."""
    from splatloc_amd import _native
    sc = make_scene(**cfg)
    b = oracle_backward(oracle_forward(sc), sc)
    _native.set_deterministic(True)
    try:
        runs = [HipRun(sc) for _ in range(3)]
    finally:
        _native.set_deterministic(False)
    names = ["means3D", "means2D", "opacities", "colors", "scales", "rotations"]
    for r in runs[1:]:
        for n in names:
            assert torch.equal(getattr(r, n).grad, getattr(runs[0], n).grad), f"{n} differs between deterministic runs"
    run = runs[0]
    kw = dict(rtol=1e-4, atol_scale=5e-6)
    assert_grad_close("dL_dmeans3D", run.np(run.means3D.grad), b["dL_dmeans3D"], **kw)
    assert_grad_close("dL_dmeans2D", run.np(run.means2D.grad), b["dL_dmeans2D"], **kw)
    assert_grad_close("dL_dopacities", run.np(run.opacities.grad), b["dL_dopacities"], **kw)
    assert_grad_close("dL_dcolors", run.np(run.colors.grad), b["dL_dcolors"], **kw)
    assert_grad_close("dL_dscales", run.np(run.scales.grad), b["dL_dscales"], **kw)
    assert_grad_close("dL_drotations", run.np(run.rotations.grad), b["dL_drotations"], **kw)
    # and the default mode still agrees with it to rounding
    free = HipRun(sc)
    for n in names:
        assert_grad_close(n, free.np(getattr(free, n).grad), run.np(getattr(run, n).grad), rtol=2e-3, atol_scale=1e-4)


@pytest.mark.parametrize("C", [4, 35])
def test_deterministic_mode_propagates_non_finite_gradients(C):
    """A NaN (and an infinity) in dL/dout must come back as a non-finite gradient in the deterministic-sum mode exactly where the
    float-atomic mode returns one: the mode meant for regression hunting may not flush the failure it should expose to 0
    (round-4 advisor finding: the NaN won the per-element maximum, every partial then scaled to 0)."""
    from splatloc_amd import _native
    sc = make_scene(3_000, 160, 96, C, 83, scale_median=0.04)
    sc.dL_dcolor[0, 40, 70] = float("nan")
    sc.dL_dcolor[min(1, C - 1), 50, 20] = float("inf")
    free = HipRun(sc)
    _native.set_deterministic(True)
    try:
        det = HipRun(sc)
    finally:
        _native.set_deterministic(False)
    for n in ("colors", "opacities", "means2D"):
        a, b = getattr(free, n).grad, getattr(det, n).grad
        bad_a, bad_b = ~torch.isfinite(a), ~torch.isfinite(b)
        assert bad_a.any(), n
        # every row the float-atomic path poisons is poisoned in the deterministic mode too (its double-precision walk may
        # turn a further inf - inf into NaN: a superset is allowed, a finite value in place of a non-finite one is not)
        assert bool((bad_b | ~bad_a).all()), f"{n}: {int((bad_a & ~bad_b).sum())} non-finite elements came back finite"
    ok = torch.isfinite(free.colors.grad).all(dim=1) & torch.isfinite(det.colors.grad).all(dim=1)
    assert ok.any() and torch.allclose(free.colors.grad[ok], det.colors.grad[ok], rtol=2e-3, atol=1e-6)
