"""The window-batched launch path (splatraster_forward_window_* / splatraster_backward_window through
splatloc_amd.rasterize_window) against V separate GaussianRasterizer calls — needs an MI355X.

SplatLoc.map renders window_size = 5 views before one backward (train_gaussians.py:195-229).  The window path runs
ONE preprocess / depth sort / tile sort keyed by (view, tile) / compositing grid for all views and sums the views'
parameter gradients in-kernel.  Contract: every per-view result — images, radii, sorted point list, tile ranges,
n_contrib, final_T — is BIT-IDENTICAL to the per-view call (and therefore, transitively, to the oracle, which the
per-view call is tested against); gradients equal the sum of the per-view gradients to float-atomic rounding, and
exactly in the deterministic-sum mode.
"""
import numpy as np
import pytest
import torch

from splatloc_amd.camera import PinholeCamera
from splatloc_amd.synthetic import make_scene
from tests.helpers import assert_grad_close, oracle_backward, oracle_forward

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _views(sc, V, dev):
    from splatloc_amd import GaussianRasterizationSettings
    cam0 = sc.camera
    W, H = cam0.image_width, cam0.image_height
    out = []
    for k in range(V):
        ang = 0.03 * (k - V // 2)
        R = torch.tensor([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], dtype=torch.float32)
        cam = PinholeCamera(W, H, cam0.fx * (1.0 + 0.02 * k), cam0.fy, cam0.cx + 0.5 * k, cam0.cy - 0.25 * k, R,
                            torch.tensor([0.02 * k, -0.01 * k, 0.05 * k])).to(dev)
        rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, sc.bg.to(dev), 1.0, cam.world_view_transform,
                                           cam.full_proj_transform, 0, cam.camera_center, False, False)
        g = tuple(torch.roll(t, shifts=11 * k + 1, dims=-1).contiguous().to(dev) for t in (sc.dL_dcolor, sc.dL_ddepth, sc.dL_dalpha))
        out.append((cam, rs, g))
    return out


def _leaves(sc, dev, cov=None):
    leaf = lambda t: t.to(dev).clone().requires_grad_(True)  # noqa: E731
    d = dict(means3D=leaf(sc.means3D), colors=leaf(sc.features), opac=leaf(sc.opacities))
    if cov is None:
        d["scales"], d["rots"], d["cov"] = leaf(sc.scales), leaf(sc.rotations), None
    else:
        d["scales"], d["rots"], d["cov"] = None, None, leaf(cov)
    return d


def _serial(sc, views, dev, cov=None, use=(True, True, True)):
    from splatloc_amd import GaussianRasterizer, introspect
    L = _leaves(sc, dev, cov)
    outs, m2s, states = [], [], []
    for cam, rs, g in views:
        m2 = torch.zeros_like(L["means3D"], requires_grad=True)
        color, depth, alpha, radii = GaussianRasterizer(raster_settings=rs)(
            means3D=L["means3D"], means2D=m2, shs=None, colors_precomp=L["colors"], opacities=L["opac"],
            scales=L["scales"], rotations=L["rots"], cov3D_precomp=L["cov"])
        fn = color.grad_fn
        sv = fn.saved_tensors
        states.append((introspect.forward_state((sv[12], sv[13], sv[14]), sc.means3D.shape[0], cam.image_width,
                                                cam.image_height, fn.num_rendered), fn.num_rendered))
        outs.append((color, depth, alpha, radii))
        m2s.append(m2)
    loss = 0
    for (color, depth, alpha, radii), (_, _, g) in zip(outs, views):
        for t, gt, u in zip((color, depth, alpha), g, use):
            if u:
                loss = loss + (t * gt).sum()
    loss.backward()
    torch.cuda.synchronize()
    return L, outs, m2s, states


def _window(sc, views, dev, cov=None, use=(True, True, True)):
    from splatloc_amd import introspect, rasterize_window
    L = _leaves(sc, dev, cov)
    m2s = [torch.zeros_like(L["means3D"], requires_grad=True) for _ in views]
    outs = rasterize_window([rs for _, rs, _ in views], L["means3D"], m2s, L["colors"], L["opac"], scales=L["scales"],
                            rotations=L["rots"], cov3D_precomp=L["cov"])
    states = []
    K = 8
    for a in range(0, len(views), K):
        fn = outs[a][0].grad_fn
        sv = fn.saved_tensors
        V = min(K, len(views) - a)
        cam = views[0][0]
        states += [(st, R) for st, R in zip(introspect.window_state((sv[7], sv[8], sv[9]), sc.means3D.shape[0], V,
                                                                   cam.image_width, cam.image_height, fn.R), fn.R)]
    loss = 0
    for (color, depth, alpha, radii), (_, _, g) in zip(outs, views):
        for t, gt, u in zip((color, depth, alpha), g, use):
            if u:
                loss = loss + (t * gt).sum()
    loss.backward()
    torch.cuda.synchronize()
    return L, outs, m2s, states


def _compare(sc, V, cov=None, use=(True, True, True), strict=False):
    dev = torch.device(DEV)
    views = _views(sc, V, dev)
    Ls, outs_s, m2_s, st_s = _serial(sc, views, dev, cov, use)
    Lw, outs_w, m2_w, st_w = _window(sc, views, dev, cov, use)
    for v in range(V):
        for a, b, nm in zip(outs_w[v], outs_s[v], ("color", "depth", "alpha", "radii")):
            assert torch.equal(a, b), f"view {v}: {nm} differs from the per-view call"
        (sw, Rw), (ss, Rs) = st_w[v], st_s[v]
        assert Rw == Rs
        for k in ("tiles_touched", "point_list", "tile_list", "ranges", "n_contrib", "final_T", "rec0", "rec1"):
            assert torch.equal(sw[k], ss[k]), f"view {v}: {k}"
        if strict:
            assert torch.equal(m2_w[v].grad, m2_s[v].grad)
        else:
            assert_grad_close(f"means2D[{v}]", m2_w[v].grad.cpu().numpy(), m2_s[v].grad.cpu().numpy())
    for k in ("means3D", "colors", "opac", "scales", "rots", "cov"):
        if Ls[k] is None:
            continue
        gw, gs = Lw[k].grad.cpu().numpy(), Ls[k].grad.cpu().numpy()
        if strict:
            # deterministic mode: the per-row sums are bit-reproducible; the window adds the V views in-kernel in view
            # order, autograd adds the V per-view tensors in backward order — float addition of V terms, reordered
            assert_grad_close(k, gw, gs, rtol=2e-6, atol_scale=1e-7)
        else:
            assert_grad_close(k, gw, gs)
    return views, Lw, outs_w, m2_w


@pytest.mark.parametrize("cfg", [
    dict(P=6000, W=320, H=240, C=4, seed=401, scale_median=0.03, V=5),     # SplatLoc's window: 5 views, [rgb | kp]
    dict(P=3000, W=333, H=201, C=35, seed=402, scale_median=0.03, V=3),    # north-star channels, ragged frame
    dict(P=4000, W=200, H=120, C=3, seed=403, scale_median=0.04, V=8),     # the maximum per launch sequence
    dict(P=2500, W=160, H=96, C=7, seed=404, scale_median=0.04, V=2),      # generic C: chunked channel passes
    dict(P=3000, W=256, H=192, C=4, seed=405, scale_median=0.03, V=1),     # a window of one view
])
def test_window_matches_per_view_calls(cfg):
    V = cfg.pop("V")
    _compare(make_scene(**cfg), V)


@pytest.mark.parametrize("mode", [0, 2])
def test_window_with_and_without_forward_teams(mode):
    """A window of small narrow frames is a launch that does not fill the machine: its longest lists may be walked by teams of
    four waves (composite_fwd.hip).  Forced on (a team for each of the 128 longest (view, tile) lists, view boundaries inside the
    launch order) and forced off, the window equals the per-view calls bit for bit, like in the default mode above."""
    from splatloc_amd import _native
    lib = _native.load()
    lib.splatraster_debug_set_fwd_team(mode)
    try:
        _compare(make_scene(P=9000, W=176, H=144, C=4, seed=411, scale_median=0.06), 3)
        _compare(make_scene(P=2000, W=64, H=64, C=3, seed=412, scale_median=0.2), 2)
    finally:
        lib.splatraster_debug_set_fwd_team(-1)


def test_window_of_4k_frames_takes_three_tile_sort_passes():
    """3840 x 2160 is 32 400 tiles per view: three views are 97 200 (view, tile) keys — 17 bits, so the tile sort runs THREE
    8-bit passes and ends in the other buffer of its ping-pong pair (every other test stays within 16 bits); one view alone is
    15 bits.  Same contract: bit-identical to the per-view calls."""
    _compare(make_scene(5000, 3840, 2160, 4, 407, scale_median=0.02), 3)


def test_window_larger_than_one_launch_sequence_is_chunked():
    _compare(make_scene(3000, 200, 120, 4, 406, scale_median=0.04), 11)    # 8 + 3


def test_window_cov3d_precomp_and_missing_aux_gradients():
    sc = make_scene(2500, 256, 192, 4, 407, scale_median=0.03)
    g = torch.Generator().manual_seed(5)
    Lm = torch.randn(2500, 3, 3, generator=g) * 0.03
    S = Lm @ Lm.transpose(1, 2)
    cov = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).contiguous()
    _compare(sc, 4, cov=cov, use=(True, False, False))     # colour-only loss: depth / alpha gradients are None


def test_window_deterministic_mode_is_reproducible():
    from splatloc_amd import _native
    sc = make_scene(5000, 320, 240, 4, 408, scale_median=0.03)
    _native.set_deterministic(True)
    try:
        _compare(sc, 5, strict=True)
        dev = torch.device(DEV)
        views = _views(sc, 5, dev)
        a = _window(sc, views, dev)[0]
        b = _window(sc, views, dev)[0]
        for k in ("means3D", "colors", "opac", "scales", "rots"):
            assert torch.equal(a[k].grad, b[k].grad), k
    finally:
        _native.set_deterministic(False)


def test_window_against_the_oracle_directly():
    """Not only transitively: every view of a 5-view window against the CPU oracle (forward bit-exact integers,
    images <= 1e-4) and the summed parameter gradients against the sum of the oracle's per-view gradients."""
    sc = make_scene(5000, 320, 240, 4, 409, scale_median=0.03)
    dev = torch.device(DEV)
    views = _views(sc, 5, dev)
    Lw, outs, m2s, states = _window(sc, views, dev)
    from oracle import oracle
    tot = {}
    for v, (cam, rs, g) in enumerate(views):
        f = oracle.forward(oracle.Settings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy), sc.bg.numpy(),
                           sc.means3D.numpy(), sc.opacities.numpy(), cam.world_view_transform.cpu().numpy(),
                           cam.full_proj_transform.cpu().numpy(), cam.camera_center.cpu().numpy(),
                           colors_precomp=sc.features.numpy(), scales=sc.scales.numpy(), rotations=sc.rotations.numpy(), omp=True)
        b = oracle.backward(f, g[0].cpu().numpy(), g[1].cpu().numpy(), g[2].cpu().numpy(), omp=True)
        st, R = states[v]
        assert R == f["num_rendered"]
        assert np.array_equal(outs[v][3].cpu().numpy(), f["radii"])
        assert np.array_equal(st["point_list"].cpu().numpy().astype(np.uint32), f["point_list"])
        assert np.array_equal(st["ranges"].cpu().numpy().astype(np.uint32), f["ranges"])
        assert np.array_equal(st["n_contrib"].cpu().numpy().astype(np.int64), f["n_contrib"].astype(np.int64))
        assert np.array_equal(st["final_T"].cpu().numpy().view(np.uint32), f["final_T"].view(np.uint32))
        assert np.abs(outs[v][0].detach().cpu().numpy() - f["color"]).max() <= 1e-4
        assert_grad_close(f"means2D[{v}]", m2s[v].grad.cpu().numpy(), b["dL_dmeans2D"])
        for k in ("dL_dmeans3D", "dL_dcolors", "dL_dopacities", "dL_dscales", "dL_drotations"):
            tot[k] = b[k].astype(np.float64) + tot.get(k, 0.0)
    for k, name in (("dL_dmeans3D", "means3D"), ("dL_dcolors", "colors"), ("dL_dopacities", "opac"),
                    ("dL_dscales", "scales"), ("dL_drotations", "rots")):
        assert_grad_close(k, Lw[name].grad.cpu().numpy(), tot[k])


def test_window_rejects_what_it_cannot_batch():
    from splatloc_amd import GaussianRasterizationSettings, rasterize_window
    sc = make_scene(100, 64, 48, 3, 410)
    dev = torch.device(DEV)
    (c0, r0, _), (c1, r1, _) = _views(sc, 2, dev)
    r1 = GaussianRasterizationSettings(*(r1[:1] + (r1.image_width + 16,) + r1[2:]))
    L = _leaves(sc, dev)
    m2 = [torch.zeros_like(L["means3D"], requires_grad=True) for _ in range(2)]
    with pytest.raises(Exception, match="share image size"):
        rasterize_window([r0, r1], L["means3D"], m2, L["colors"], L["opac"], scales=L["scales"], rotations=L["rots"])
    with pytest.raises(Exception, match="precomputed colors"):
        rasterize_window([r0], L["means3D"], m2[:1], None, L["opac"], scales=L["scales"], rotations=L["rots"])


def test_window_empty_scene_and_all_culled():
    from splatloc_amd import rasterize_window
    dev = torch.device(DEV)
    sc = make_scene(400, 128, 96, 3, 411)
    sc.means3D[:, 2] = -sc.means3D[:, 2]
    sc.bg = torch.tensor([0.2, 0.4, 0.6])
    views = _views(sc, 3, dev)
    L, outs, m2s, states = _window(sc, views, dev)
    for (color, depth, alpha, radii), (st, R) in zip(outs, states):
        assert R == 0 and int(radii.abs().sum()) == 0
        assert torch.allclose(color[1], torch.full_like(color[1], 0.4)) and float(alpha.abs().max()) == 0.0
    assert float(L["means3D"].grad.abs().sum()) == 0.0
    e = lambda *s: torch.zeros(*s, device=dev, requires_grad=True)  # noqa: E731
    outs = rasterize_window([rs for _, rs, _ in views], e(0, 3), [e(0, 3) for _ in views], e(0, 3), e(0, 1), scales=e(0, 3),
                            rotations=e(0, 4))
    assert outs[2][0].shape == (3, 96, 128) and outs[2][3].numel() == 0
    sum(o[0].sum() for o in outs).backward()


def test_window_soak_random_shapes():
    """Random window sizes / scene sizes / frame sizes / channel counts back to back: every per-view result of the window
    bit-identical to the per-view call (sort path switches with V * P, scan look-back under different block counts, the
    16-bit (view, tile) keys near their pass boundaries, buffer re-sizing between windows)."""
    g = torch.Generator().manual_seed(2024)
    for it in range(16):
        V = int(torch.randint(1, 9, (1,), generator=g).item())
        P = int(10 ** (1.0 + 3.7 * torch.rand(1, generator=g).item()))
        W = int(16 + torch.randint(0, 700, (1,), generator=g).item())
        H = int(16 + torch.randint(0, 500, (1,), generator=g).item())
        C = [1, 3, 4, 7, 35][int(torch.randint(0, 5, (1,), generator=g).item())]
        sm = 10 ** (-2.3 + 1.2 * torch.rand(1, generator=g).item())
        _compare(make_scene(P, W, H, C, seed=700 + it, scale_median=sm), V)


# Full-size gradient bars (round 4; measured distributions: profiles/r04_grad_bars.json, tools/grad_bar_probe.py).
# tensor-scale bar: |d| <= rtol |ref| + atol_scale max|tensor|;  row bar: |d| <= rtol |ref| + row_atol max|row| (helpers).
FULL_TENSOR = dict(rtol=1e-4, atol_scale=2e-5)      # accurate (deterministic) mode (round 3: 2e-3 / 1e-4)
FULL_TENSOR_ATOMIC = dict(rtol=1e-4, atol_scale=5e-5)   # normal path: float atomics reorder the sums from run to run
NAMES = (("dL_dmeans3D", "means3D"), ("dL_dcolors", "colors"), ("dL_dopacities", "opac"), ("dL_dscales", "scales"),
         ("dL_drotations", "rots"))


def _rows_strict(tag, got, tot):
    """per-ROW bars of the full-size tests: rtol 1e-4 + 1e-3 of the row's own maximum (colours: 1e-5); the one-element opacity rows
    are a purely relative bar on a sum of signed G dL/dalpha terms that cancels for some Gaussians"""
    from tests.helpers import assert_grad_rows_close
    for k, nm in NAMES:
        if k == "dL_dcolors":
            assert_grad_rows_close(f"{tag} {k}", got[nm], tot[k], rtol=1e-4, row_atol=1e-5)
        elif k == "dL_dopacities":
            assert_grad_rows_close(f"{tag} {k}", got[nm], tot[k], rtol=1e-4, row_atol=1e-3, allow_frac=1e-3, outlier_factor=float("inf"))
        else:
            assert_grad_rows_close(f"{tag} {k}", got[nm], tot[k], rtol=1e-4, row_atol=1e-3, allow_frac=1e-4, outlier_factor=30.0)


def _oracle_window(sc, views, mode):
    from oracle import oracle
    oracle.set_alpha_mode(mode)
    try:
        tot, per_view = {}, []
        for cam, rs, g in views:
            f = oracle.forward(oracle.Settings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy), sc.bg.numpy(),
                               sc.means3D.numpy(), sc.opacities.numpy(), cam.world_view_transform.cpu().numpy(),
                               cam.full_proj_transform.cpu().numpy(), cam.camera_center.cpu().numpy(),
                               colors_precomp=sc.features.numpy(), scales=sc.scales.numpy(), rotations=sc.rotations.numpy(), omp=True)
            b = oracle.backward(f, g[0].cpu().numpy(), g[1].cpu().numpy(), g[2].cpu().numpy(), omp=True)
            for k, _ in NAMES:
                tot[k] = b[k].astype(np.float64) + tot.get(k, 0.0)
            keep = {k: f[k] for k in ("num_rendered", "radii", "point_list", "ranges", "n_contrib", "final_T", "color", "depth")}
            keep["dL_dmeans2D"] = b["dL_dmeans2D"]
            per_view.append(keep)
            del f, b
    finally:
        oracle.set_alpha_mode(0)
    return tot, per_view


@pytest.mark.parametrize("name,V", [("S2", 3), ("S2-ref-layout", 5)])
def test_window_full_size_against_oracle(name, V):
    """BASELINE.json's full sizes through the window path: 3 views of S2 (500k Gaussians, 1920x1080, C = 35) and the
    5-view window SplatLoc really renders (500k, 640x480, C = 4) — every view's radii / point list / ranges / n_contrib /
    final_T bit-exact against the CPU oracle, images <= 1e-4, per-view dL/dmeans2D and the SUMMED parameter gradients
    against the sum of the oracle's per-view gradients:
      * tensor bar: rtol 1e-4 + 5e-5 of the tensor's scale (2e-5 in the deterministic mode; round 3: 2e-3 + 1e-4);
      * per-ROW bar: rtol 1e-4 + 1e-3 of the row's own maximum for every tensor, so that a Gaussian whose gradient is a
        thousand times smaller than the largest cannot hide —
    for BOTH the normal path (float atomics; since round 4 the backward walks back to front) and the
    deterministic / accurate mode."""
    from splatloc_amd import _native
    from splatloc_amd.synthetic import make_workload
    from tests.helpers import assert_grad_rows_close
    sc = make_workload(name)
    dev = torch.device(DEV)
    views = _views(sc, V, dev)
    tot, per_view = _oracle_window(sc, views, 0)
    Lw, outs, m2s, states = _window(sc, views, dev)
    for v, f in enumerate(per_view):
        st, R = states[v]
        assert R == f["num_rendered"] > 500_000
        assert np.array_equal(outs[v][3].cpu().numpy(), f["radii"])
        assert np.array_equal(st["point_list"].cpu().numpy().astype(np.uint32), f["point_list"])
        assert np.array_equal(st["ranges"].cpu().numpy().astype(np.uint32), f["ranges"])
        assert np.array_equal(st["n_contrib"].cpu().numpy().astype(np.int64), f["n_contrib"].astype(np.int64))
        assert np.array_equal(st["final_T"].cpu().numpy().view(np.uint32), f["final_T"].view(np.uint32))
        assert np.abs(outs[v][0].detach().cpu().numpy() - f["color"]).max() <= 1e-4
        assert np.abs(outs[v][1].detach().cpu().numpy() - f["depth"]).max() <= 1e-4 * max(1.0, float(f["depth"].max()))
        assert_grad_close(f"means2D[{v}]", m2s[v].grad.cpu().numpy(), f["dL_dmeans2D"], **FULL_TENSOR_ATOMIC)
    for k, nm in NAMES:
        assert_grad_close(k, Lw[nm].grad.cpu().numpy(), tot[k], **FULL_TENSOR_ATOMIC)
    # the normal path per ROW, since the backward walks back to front (round 4): the same strict bars as the accurate mode
    _rows_strict("rows", {nm: Lw[nm].grad.cpu().numpy() for _, nm in NAMES}, tot)
    for v, f in enumerate(per_view):
        assert_grad_rows_close(f"rows means2D[{v}]", m2s[v].grad.cpu().numpy(), f["dL_dmeans2D"], rtol=1e-4, row_atol=1e-3,
                               allow_frac=1e-4, outlier_factor=30.0)
    del Lw, outs, m2s, states
    # ---- deterministic / accurate mode: tensor bar + strict per-row bars ----
    _native.set_deterministic(True)
    try:
        Ld, outs_d, m2d, _ = _window(sc, views, dev)
    finally:
        _native.set_deterministic(False)
    for v, f in enumerate(per_view):
        assert_grad_close(f"det means2D[{v}]", m2d[v].grad.cpu().numpy(), f["dL_dmeans2D"], **FULL_TENSOR)
        assert_grad_rows_close(f"det rows means2D[{v}]", m2d[v].grad.cpu().numpy(), f["dL_dmeans2D"], rtol=1e-4, row_atol=1e-3,
                               allow_frac=1e-4, outlier_factor=30.0)
    for k, nm in NAMES:
        assert_grad_close("det " + k, Ld[nm].grad.cpu().numpy(), tot[k], **FULL_TENSOR)
    _rows_strict("det rows", {nm: Ld[nm].grad.cpu().numpy() for _, nm in NAMES}, tot)
    # ---- and against the SPEC: oracle mode 1, the lineage's literal exp form — identical except where an
    # alpha >= 1/255 / T < 1e-4 decision flips within rounding: a counted fraction of elements, bounded in size ----
    tot1, per_view1 = _oracle_window(sc, views, 1)
    for v, f in enumerate(per_view1):
        assert np.array_equal(outs_d[v][3].cpu().numpy(), f["radii"])
        flipped = outs_d[v][0].detach().cpu().numpy() - f["color"]
        assert (np.abs(flipped).max(axis=0) > 1e-4).mean() <= 1e-3
        assert_grad_close(f"mode1 means2D[{v}]", m2d[v].grad.cpu().numpy(), f["dL_dmeans2D"], allow_frac=1e-3, outlier_factor=30.0,
                          **FULL_TENSOR)
    for k, nm in NAMES:
        assert_grad_close("mode1 " + k, Ld[nm].grad.cpu().numpy(), tot1[k], allow_frac=1e-3, outlier_factor=30.0, **FULL_TENSOR)


def test_head_and_last_outputs_of_a_wide_table():
    """[rgb | 31 feature channels | kp_score] (C = 35): image[:3] and image[-1] as separate autograd outputs
    (split_last = 3) give the images and gradients of slicing the one 35-channel output — without the zero-padded
    [C,H,W] gradients; with only the head reaching the loss the backward runs on 3 channels."""
    from splatloc_amd import rasterize_window, _native
    sc = make_scene(1500, 160, 96, 35, seed=11, scale_median=0.1)
    views = _views(sc, 2, DEV)
    names = ("means3D", "colors", "opac", "scales", "rots")
    _native.set_deterministic(True)
    try:
        res = {}
        for mode in ("slice", "split", "slice_head", "split_head"):
            L = _leaves(sc, DEV)
            m2s = [torch.zeros_like(L["means3D"], requires_grad=True) for _ in views]
            split = 3 if mode.startswith("split") else False
            outs = rasterize_window([rs for _, rs, _ in views], L["means3D"], m2s, L["colors"], L["opac"], scales=L["scales"],
                                    rotations=L["rots"], split_last=split)
            loss = 0
            imgs = []
            for o, (_, _, g) in zip(outs, views):
                if split:
                    rgb, last, depth, alpha, _ = o
                else:
                    img, depth, alpha, _ = o
                    rgb, last = img[:3], img[-1]
                imgs.append((rgb.detach().clone(), last.detach().clone()))
                loss = loss + (rgb * g[0][:3]).sum() + (depth * g[1]).sum()
                if not mode.endswith("head"):
                    loss = loss + (last * g[0][-1]).sum()
            loss.backward()
            res[mode] = (imgs, {n: L[n].grad.clone() for n in names}, [m.grad.clone() for m in m2s])
    finally:
        _native.set_deterministic(False)
    for a, b in (("slice", "split"), ("slice_head", "split_head")):
        for (r0, l0), (r1, l1) in zip(res[a][0], res[b][0]):
            assert torch.equal(r0, r1) and torch.equal(l0, l1)
        for n in names:
            assert_grad_close(f"{b} {n}", res[b][1][n].cpu().numpy(), res[a][1][n].cpu().numpy(), rtol=2e-5, atol_scale=2e-6)
        for g0, g1 in zip(res[a][2], res[b][2]):
            assert_grad_close(f"{b} means2D", g1.cpu().numpy(), g0.cpu().numpy(), rtol=2e-5, atol_scale=2e-6)
    # feature columns 3..33 received no gradient in any mode; column 34 only when kp_score reached the loss
    assert float(res["split"][1]["colors"][:, 3:34].abs().max()) == 0.0
    assert float(res["split_head"][1]["colors"][:, 3:].abs().max()) == 0.0


def test_window_chunks_respect_the_row_limit():
    """A window's (view, Gaussian) rows are addressed with 24 bits: 8 views of a scene of more than 2^21 Gaussians go out as
    two launch sequences (7 + 1 views) instead of failing — same images as the per-view calls."""
    from splatloc_amd import GaussianRasterizer, rasterize_window
    P = (1 << 21) + 5000
    sc = make_scene(20000, 96, 64, 3, seed=5, scale_median=0.05)
    rep = (P + 19999) // 20000
    big = lambda t: t.repeat((rep,) + (1,) * (t.dim() - 1))[:P].contiguous().to(DEV)  # noqa: E731
    m3, col, opa, sca, rot = (big(t) for t in (sc.means3D, sc.features, sc.opacities, sc.scales, sc.rotations))
    m3 = m3 + 0.001 * torch.arange(P, device=DEV, dtype=torch.float32)[:, None] / P       # no exact duplicates
    views = _views(sc, 8, DEV)
    with torch.no_grad():
        outs = rasterize_window([rs for _, rs, _ in views], m3, [torch.zeros_like(m3) for _ in views], col, opa,
                                scales=sca, rotations=rot)
        for (_, rs, _), o in zip(views[::3], outs[::3]):
            c, d, a, r = GaussianRasterizer(raster_settings=rs)(means3D=m3, means2D=torch.zeros_like(m3), shs=None,
                                                                colors_precomp=col, opacities=opa, scales=sca, rotations=rot,
                                                                cov3D_precomp=None)
            assert torch.equal(c, o[0]) and torch.equal(d, o[1]) and torch.equal(a, o[2]) and torch.equal(r, o[3])
