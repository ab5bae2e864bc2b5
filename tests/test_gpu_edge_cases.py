"""Edge cases of the HIP path against the CPU oracle (needs an MI355X): tiny and ragged
images, screen-filling Gaussians, exact depth ties, single Gaussian, saturated / vanishing
opacity, odd channel counts, very deep per-tile lists.  Same bar as test_gpu_parity.py.
"""
import numpy as np
import pytest
import torch

from splatloc_amd.synthetic import make_scene
from tests.helpers import HipRun, assert_grad_close, assert_grad_rows_close, oracle_backward, oracle_forward
from tests.test_gpu_parity import _check_backward, _check_forward

pytestmark = pytest.mark.gpu


def _run_and_check(sc, backward=True):
    f = oracle_forward(sc)
    run = HipRun(sc, backward=backward)
    _check_forward(run, f, sc)
    if backward:
        _check_backward(run, oracle_backward(f, sc))
    return run, f


@pytest.mark.parametrize("W,H", [(7, 5), (16, 16), (17, 9), (40, 24), (8, 64)])
def test_tiny_and_ragged_images(W, H):
    """Images smaller than a tile / a quadrant, one-pixel overhangs."""
    sc = make_scene(300, W, H, 4, seed=100 + W, scale_median=0.08)
    _run_and_check(sc)


def test_screen_filling_gaussians():
    """Radii far larger than the image: every Gaussian lands in every tile."""
    sc = make_scene(200, 96, 64, 3, seed=41, scale_median=1.5)
    run, f = _run_and_check(sc)
    tiles = ((96 + 15) // 16) * ((64 + 15) // 16)
    assert (f["tiles_touched"][f["radii"] > 0] == tiles).mean() > 0.5


def test_gaussians_over_more_than_255_tiles():
    """The depth-order words carry min(tiles_touched, 255) for the offsets scan; a Gaussian over >= 255 tiles takes the
    look-up path (scan_sort.hip perm_value).  Mixed with small ones so that both paths meet in one scan."""
    big = make_scene(60, 320, 240, 3, seed=43, scale_median=2.5)
    small = make_scene(400, 320, 240, 3, seed=44, scale_median=0.05)
    sc = big
    for name in ("means3D", "features", "opacities", "scales", "rotations"):
        setattr(sc, name, torch.cat([getattr(big, name), getattr(small, name)], 0))
    run, f = _run_and_check(sc)
    tt = f["tiles_touched"]
    assert (tt >= 255).sum() >= 10 and ((tt > 0) & (tt < 255)).sum() >= 50


def test_exact_depth_ties_keep_index_order():
    """Identical depths: the (tile, depth, index) order must fall back to the Gaussian index."""
    sc = make_scene(600, 128, 96, 3, seed=42, scale_median=0.06)
    sc.means3D[:, 2] = torch.round(sc.means3D[:, 2] * 2.0) / 2.0     # 12 distinct depths
    sc.means3D[300:] = sc.means3D[:300]                              # exact duplicates as well
    run, f = _run_and_check(sc)
    pl = f["point_list"].astype(np.int64)
    z = f["view_depth"]
    r = f["ranges"].astype(np.int64)
    t = int(np.argmax(r[:, 1] - r[:, 0]))
    seg = pl[r[t, 0]:r[t, 1]]
    same = z[seg[1:]] == z[seg[:-1]]
    assert same.any() and (seg[1:][same] > seg[:-1][same]).all()


def test_single_gaussian():
    sc = make_scene(1, 64, 48, 3, seed=43, scale_median=0.2)
    sc.means3D[0] = torch.tensor([0.0, 0.0, 2.0])
    run, f = _run_and_check(sc)
    assert run.num_rendered > 0


def test_saturated_and_vanishing_opacity():
    """opacity = 1 (alpha clamps at 0.99), opacity below 1/255 (never contributes), and 0."""
    sc = make_scene(900, 128, 96, 4, seed=44, scale_median=0.06)
    sc.opacities[:300] = 1.0
    sc.opacities[300:600] = 1.0 / 512.0
    sc.opacities[600:650] = 0.0
    run, f = _run_and_check(sc)
    g = run.np(run.opacities.grad)
    assert np.all(np.isfinite(g))
    assert np.all(run.np(run.colors.grad)[300:650] == 0.0)           # never composited


@pytest.mark.parametrize("C", [2, 5, 6, 9, 16, 17, 32, 33, 48, 67])
def test_channel_counts(C):
    """Specialised kernels (1, 2, 3, 4, 8, 16, 32, 35) and the chunked generic path."""
    sc = make_scene(1200, 112, 80, C, seed=50 + C, scale_median=0.05)
    _run_and_check(sc)


def test_very_deep_lists():
    """Hundreds of overlapping Gaussians per pixel: more than one staging round per 64-entry
    chunk in every quadrant, early termination (T < 1e-4) in most pixels."""
    sc = make_scene(6000, 96, 96, 35, seed=45, scale_median=0.3)
    run, f = _run_and_check(sc)
    assert float(f["final_T"].mean()) < 0.05
    r = f["ranges"].astype(np.int64)
    assert (r[:, 1] - r[:, 0]).min() > 1000


def test_background_shorter_than_channels():
    """SplatLoc passes a 3-entry background with C = 4: missing entries are 0."""
    sc = make_scene(800, 96, 64, 4, seed=46, scale_median=0.03)
    sc.bg = torch.tensor([0.3, 0.6, 0.9])
    run, f = _run_and_check(sc)
    empty = f["final_T"] == 1.0
    assert empty.any()
    col = run.np(run.color)
    assert np.allclose(col[:3][:, empty], np.array([[0.3], [0.6], [0.9]]), atol=1e-6)
    assert np.all(col[3][empty] == 0.0)


def test_everything_behind_or_beside_the_camera():
    sc = make_scene(500, 64, 48, 3, seed=47, scale_median=0.02)
    sc.means3D[:250, 2] = -sc.means3D[:250, 2]           # behind
    sc.means3D[250:, 0] += 1000.0                        # far outside the frustum
    run, f = _run_and_check(sc)
    assert run.num_rendered == 0 and int(run.radii.abs().sum()) == 0
    for t in (run.means3D, run.colors, run.opacities, run.scales, run.rotations):
        assert float(t.grad.abs().sum()) == 0.0


def test_sizes_change_between_calls_in_one_process():
    """Densification changes P every few iterations and key-frames differ in size: every call
    sizes its own buffers (the host landing slot of the instance count grows on demand)."""
    for P, W, H, C, sm in ((500, 64, 48, 4, 0.05), (70_000, 320, 240, 4, 0.01), (3, 17, 9, 3, 0.2),
                           (1_200_000, 96, 64, 3, 0.002), (9_000, 200, 120, 35, 0.03)):
        sc = make_scene(P, W, H, C, seed=P % 97, scale_median=sm)
        _run_and_check(sc, backward=(P < 100_000))


def test_two_streams_interleaved():
    """Frames enqueued on two different HIP streams from one host thread: results equal the
    single-stream ones (the C ABI only ever touches the stream it is given)."""
    sc_a = make_scene(4000, 160, 96, 4, seed=61, scale_median=0.04)
    sc_b = make_scene(6000, 128, 128, 35, seed=62, scale_median=0.03)
    ref_a, ref_b = HipRun(sc_a), HipRun(sc_b)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    runs = {}
    for rep in range(3):
        with torch.cuda.stream(s1):
            runs["a"] = HipRun(sc_a)
        with torch.cuda.stream(s2):
            runs["b"] = HipRun(sc_b)
    torch.cuda.synchronize()
    for k, ref in (("a", ref_a), ("b", ref_b)):
        r = runs[k]
        assert torch.equal(r.radii, ref.radii) and r.num_rendered == ref.num_rendered
        assert float((r.color - ref.color).abs().max()) == 0.0
        assert float((r.depth - ref.depth).abs().max()) == 0.0
        scale = float(ref.means3D.grad.abs().max())
        assert float((r.means3D.grad - ref.means3D.grad).abs().max()) <= 2e-3 * scale


def test_forward_under_no_grad_like_eval_rendering():
    """utils/eval_utils.py:46 renders inside torch.no_grad(): same image, no autograd state."""
    from splatloc_amd import GaussianRasterizer
    from tests.helpers import hip_settings
    sc = make_scene(5000, 320, 240, 4, seed=71, scale_median=0.03)
    ref = HipRun(sc, backward=False)
    dev = torch.device("cuda:0")
    s = sc.to(dev)
    with torch.no_grad():
        color, depth, alpha, radii = GaussianRasterizer(raster_settings=hip_settings(s, dev))(
            means3D=s.means3D, means2D=torch.zeros_like(s.means3D), shs=None, colors_precomp=s.features,
            opacities=s.opacities, scales=s.scales, rotations=s.rotations, cov3D_precomp=None)
    assert color.grad_fn is None and not color.requires_grad
    assert torch.equal(color, ref.color.detach()) and torch.equal(depth, ref.depth.detach())
    assert torch.equal(alpha, ref.alpha.detach()) and torch.equal(radii, ref.radii)


def test_lookback_stall_is_reported_and_never_wrong():
    """A decoupled look-back (one-pass scan / one-sweep radix sort) that waits longer than its SOFT spin bound
    keeps waiting, so the frame stays bit-exact and the forward call still succeeds; the stall surfaces only
    through splatraster_poll_errors() as the non-fatal SPLATRASTER_WARN_LOOKBACK_STALL.  With the bound forced
    to 0 every block that has to wait at all raises the host-mapped flag."""
    from splatloc_amd import _native
    lib = _native.load()
    sc = make_scene(300_000, 640, 480, 3, seed=71, scale_median=0.006)   # 74 sort blocks, 147 scan tiles
    f = oracle_forward(sc)
    assert lib.splatraster_poll_errors() in (0, _native.WARN_LOOKBACK_STALL)   # drain
    assert lib.splatraster_poll_errors() == 0
    _native.check(lib.splatraster_debug_set_spin_limit(0), "set_spin_limit")
    try:
        warned = False
        for _ in range(4):                       # whether a block waits is timing dependent: a few frames make it certain
            run = HipRun(sc, backward=False)     # never raises: a late frame is not a failed frame
            torch.cuda.synchronize()
            assert np.array_equal(run.np(run.state["point_list"]).astype(np.uint32), f["point_list"])
            if _native.poll_stall():
                warned = True
                break
        assert warned, "no look-back had to wait in 4 frames of 300k Gaussians?"
    finally:
        _native.check(lib.splatraster_debug_set_spin_limit(1 << 24), "restore spin limit")
        torch.cuda.synchronize()
        lib.splatraster_poll_errors()            # drain flags raised by frames still in flight
    run = HipRun(sc, backward=False)             # healthy again, bit-exact
    _check_forward(run, f, sc)


def test_empty_scene_backward_zeroes_pose_gradients():
    """P = 0 with camera tensors that require grad: dL/dviewmatrix / projmatrix / campos are defined zeros
    (the backward returns early; the buffers must not be left uninitialised)."""
    from splatloc_amd import GaussianRasterizationSettings, GaussianRasterizer
    sc = make_scene(10, 64, 48, 3, seed=5)
    dev = torch.device("cuda:0")
    cam = sc.camera
    V = cam.world_view_transform.to(dev).clone().requires_grad_(True)
    PM = cam.full_proj_transform.to(dev).clone().requires_grad_(True)
    cp = cam.camera_center.to(dev).clone().requires_grad_(True)
    # poison the allocator's free blocks so that an un-zeroed buffer would show
    junk = [torch.full((64,), float("nan"), device=dev) for _ in range(64)]
    del junk
    rs = GaussianRasterizationSettings(48, 64, cam.tanfovx, cam.tanfovy, torch.ones(3, device=dev), 1.0, V, PM, 0, cp,
                                       False, False)
    e = lambda *s: torch.zeros(*s, device=dev, requires_grad=True)  # noqa: E731
    color, depth, alpha, radii = GaussianRasterizer(raster_settings=rs)(
        means3D=e(0, 3), means2D=e(0, 3), shs=None, colors_precomp=e(0, 3), opacities=e(0, 1), scales=e(0, 3),
        rotations=e(0, 4), cov3D_precomp=None)
    (color.sum() + depth.sum()).backward()
    for t in (V, PM, cp):
        assert t.grad is not None and bool((t.grad == 0).all())


@pytest.mark.parametrize("C,aux", [(4, True), (3, True), (4, False), (1, True)])
def test_split_backward_matches_unsplit(C, aux):
    """One small frame of a narrow layout runs the backward as FOUR waves per quadrant, one per quarter of the tile's list; a
    wave whose quarter ends in front of a pixel's last contributor starts from the boundary state rebuilt from the forward's
    segment records (T_b, A_b = what the later quarters contribute / T_b; common.h).  Same images bit for bit (the forward's
    image arithmetic is untouched) and the same gradients as the one-wave-per-quadrant backward per gradient ROW, on deep lists
    (hundreds of entries per tile).  (The deterministic mode never splits: both of its runs are the un-split kernel.)"""
    from splatloc_amd import _native
    lib = _native.load()
    sc = make_scene(40_000, 256, 256, C, seed=90 + C, scale_median=0.05)
    names = ["means3D", "means2D", "opacities", "colors", "scales", "rotations"]
    for det in (False, True):
        _native.set_deterministic(det)
        try:
            lib.splatraster_debug_set_split_max_waves(0)
            a = HipRun(sc, use_depth=aux, use_alpha=aux)
            lib.splatraster_debug_set_split_max_waves(-1)
            b = HipRun(sc, use_depth=aux, use_alpha=aux)
        finally:
            lib.splatraster_debug_set_split_max_waves(-1)
            _native.set_deterministic(False)
        rng = a.state["ranges"].long()
        assert int((rng[:, 1] - rng[:, 0]).max()) >= 4 * 256          # lists long enough to be split in four
        assert torch.equal(a.color, b.color) and torch.equal(a.depth, b.depth) and torch.equal(a.alpha, b.alpha)
        assert torch.equal(a.state["n_contrib"], b.state["n_contrib"])
        for n in names:
            ga, gb = getattr(a, n).grad.cpu().numpy(), getattr(b, n).grad.cpu().numpy()
            if det:
                assert_grad_close(n, gb, ga, rtol=2e-5, atol_scale=2e-6)
            else:
                assert_grad_close(n, gb, ga, rtol=1e-4, atol_scale=2e-5)
                assert_grad_rows_close("rows " + n, gb, ga, rtol=1e-4, row_atol=1e-3, allow_frac=1e-3, outlier_factor=100.0)
    f = oracle_forward(sc)
    bo = oracle_backward(f, sc, use_depth=aux, use_alpha=aux)
    assert_grad_close("dL_dmeans3D vs oracle", b.np(b.means3D.grad), bo["dL_dmeans3D"])
    assert_grad_close("dL_dcolors vs oracle", b.np(b.colors.grad), bo["dL_dcolors"])


def _clustered(P, W, H, C, seed, clusters):
    """a scene whose first Gaussians are moved onto a few pixels (distinct depths): tile lists of chosen lengths"""
    sc = make_scene(P, W, H, C, seed=seed, scale_median=0.02)
    g = torch.Generator().manual_seed(seed + 1)
    cam = sc.camera
    fx, fy = cam.image_width / (2.0 * cam.tanfovx), cam.image_height / (2.0 * cam.tanfovy)
    at = 0
    for n, (px, py) in clusters:
        z = 1.0 + 4.0 * torch.rand(n, generator=g)
        sc.means3D[at:at + n, 0] = (px - (W - 1) / 2.0) / fx * z
        sc.means3D[at:at + n, 1] = (py - (H - 1) / 2.0) / fy * z
        sc.means3D[at:at + n, 2] = z
        sc.scales[at:at + n] = 0.004
        sc.opacities[at:at + n] = 0.02 + 0.05 * torch.rand(n, 1, generator=g)      # translucent: the whole list contributes
        at += n
    return sc


@pytest.mark.parametrize("front_end", [0, 1])
@pytest.mark.parametrize("C,aux", [(4, True), (3, False)])
def test_split_backward_with_8_and_16_parts_matches_unsplit(C, aux, front_end):
    """Round 6: the longest lists of a split launch get 8 parts (from 1 024 entries) or 16 (from 2 048), walked by extra workgroups
    at the head of the backward's grid; the part count is written with the launch order and read by the forward (segment records)
    and the backward.  Lists of ~700, ~1 500, ~2 600 and ~5 000 entries in one frame (4 / 8 / 16 / 16 parts), both front ends:
    bit-identical images, gradients per row equal to the one-wave-per-quadrant backward's, and the oracle's."""
    from splatloc_amd import _native
    lib = _native.load()
    sc = _clustered(30_000, 256, 192, C, seed=300 + C, clusters=[(5_000, (40.3, 30.2)), (2_500, (120.7, 100.1)), (1_400, (200.2, 50.6)),
                                                                   (600, (90.5, 150.4))])
    names = ["means3D", "means2D", "opacities", "colors", "scales", "rotations"]
    _native.set_front_end(front_end)
    try:
        lib.splatraster_debug_set_split_max_waves(0)
        a = HipRun(sc, use_depth=aux, use_alpha=aux)
        lib.splatraster_debug_set_split_max_waves(-1)
        b = HipRun(sc, use_depth=aux, use_alpha=aux)
    finally:
        lib.splatraster_debug_set_split_max_waves(-1)
        _native.set_front_end(-1)
    lens = (a.state["ranges"][:, 1] - a.state["ranges"][:, 0]).long()
    assert int((lens >= 2048).sum()) >= 2 and int(((lens >= 1024) & (lens < 2048)).sum()) >= 1 and int(lens.max()) >= 4096
    assert torch.equal(a.color, b.color) and torch.equal(a.depth, b.depth) and torch.equal(a.alpha, b.alpha)
    assert torch.equal(a.state["n_contrib"], b.state["n_contrib"])
    # the clusters' pixels really are deep: some pixel has more than 2 048 contributors
    assert int(a.state["n_contrib"].max()) > 2048
    f = oracle_forward(sc)
    bo = oracle_backward(f, sc, use_depth=aux, use_alpha=aux)
    for n in names:
        ga, gb = getattr(a, n).grad.cpu().numpy(), getattr(b, n).grad.cpu().numpy()
        assert_grad_close(n, gb, ga, rtol=1e-4, atol_scale=2e-5)
        # (a [P, 1] tensor makes the row bar a purely RELATIVE one: behind 2 000 translucent contributors dL/dopacity is a cancelling
        #  sum ten orders of magnitude below the tensor's maximum — such rows are counted, not bounded, and held against the oracle below)
        one = ga.reshape(ga.shape[0], -1).shape[1] == 1
        assert_grad_rows_close("rows " + n, gb, ga, rtol=1e-4, row_atol=1e-3, allow_frac=1e-3, outlier_factor=float("inf") if one else 100.0)
    assert_grad_close("dL_dmeans3D vs oracle", b.np(b.means3D.grad), bo["dL_dmeans3D"])
    assert_grad_close("dL_dcolors vs oracle", b.np(b.colors.grad), bo["dL_dcolors"])
    assert_grad_close("dL_dopacities vs oracle", b.np(b.opacities.grad), bo["dL_dopacities"])
    # where split and unsplit disagree by more than 1e-3 of the value, the SPLIT backward is not the one further from the oracle (its
    # parts restart from exact boundary states; the one-wave walk carries 5 000 steps of rounding)
    go = bo["dL_dopacities"].ravel().astype(np.float64)
    ua, sb = a.np(a.opacities.grad).ravel().astype(np.float64), b.np(b.opacities.grad).ravel().astype(np.float64)
    far = np.abs(ua - sb) > 1e-3 * np.abs(go)
    assert far.sum() <= 30 and (np.abs(sb[far] - go[far]) <= np.abs(ua[far] - go[far]) + 1e-3 * np.abs(go[far])).mean() >= 0.6 if far.any() else True


@pytest.mark.parametrize("C,W,H,P,scale", [(4, 256, 256, 40_000, 0.05), (3, 250, 130, 6_000, 0.12), (1, 64, 48, 3_000, 0.3), (2, 16, 16, 700, 0.25)])
def test_team_forward_changes_nothing(C, W, H, P, scale):
    """Narrow layouts on a frame that does not fill the machine run the forward with a TEAM of four waves per quadrant (two
    evaluate alpha for a step of candidates into LDS, one runs the transmittance chain a step behind and leaves the weights,
    one accumulates colours / depth / segment sums another step behind; composite_fwd.hip).
    Same operations in the same order as the one-wave kernel: images, depth, alpha, final_T, n_contrib and the segment
    records the split backward starts from (hence its gradients, in the deterministic-free default mode up to the atomics'
    order; here compared in the deterministic mode, which does not split, bit for bit) — on deep lists (several steps and
    chunks per tile), ragged frames, steps with an odd number of candidates and frames with a single tile."""
    from splatloc_amd import _native
    lib = _native.load()
    sc = make_scene(P, W, H, C, seed=300 + C, scale_median=scale)
    runs = {}
    try:
        for mode in (0, 2):     # one wave per quadrant | a team for each of the 128 longest lists, one workgroup per other tile
            lib.splatraster_debug_set_fwd_team(mode)
            _native.set_deterministic(True)
            runs[mode, "det"] = HipRun(sc)
            _native.set_deterministic(False)
            runs[mode, "fast"] = HipRun(sc)
    finally:
        lib.splatraster_debug_set_fwd_team(-1)
        _native.set_deterministic(False)
    for kind in ("det", "fast"):
        a, b = runs[0, kind], runs[2, kind]
        assert torch.equal(a.color, b.color) and torch.equal(a.depth, b.depth) and torch.equal(a.alpha, b.alpha), kind
        assert torch.equal(a.state["n_contrib"], b.state["n_contrib"]) and torch.equal(a.state["final_T"], b.state["final_T"])
    a, b = runs[0, "det"], runs[2, "det"]
    for n in ("means3D", "means2D", "opacities", "colors", "scales", "rotations"):
        assert torch.equal(getattr(a, n).grad, getattr(b, n).grad), n
    a, b = runs[0, "fast"], runs[2, "fast"]     # (split backward from the segment records of either forward)
    for n in ("means3D", "opacities", "colors", "scales", "rotations"):
        assert_grad_close(n, getattr(b, n).grad.cpu().numpy(), getattr(a, n).grad.cpu().numpy(), rtol=1e-4, atol_scale=2e-5)
    rng = a.state["ranges"].long()
    if P >= 40_000:
        assert int((rng[:, 1] - rng[:, 0]).max()) >= 4 * 256


def test_team_forward_full_size():
    """The three forwards of a narrow single frame at BASELINE's reference layout (500k Gaussians, 640x480, C = 4: 4 800 quadrant
    lists of ~1 000 entries): one wave per quadrant, the default (teams for the lists that stand out) and a team for each of the
    128 longest lists — identical images, depth, alpha, final_T and n_contrib."""
    from splatloc_amd import _native
    from splatloc_amd.synthetic import make_workload
    lib = _native.load()
    sc = make_workload("S2-ref-layout")
    outs = {}
    try:
        for mode in (0, -1, 2):
            lib.splatraster_debug_set_fwd_team(mode)
            r = HipRun(sc, backward=False)
            outs[mode] = (r.color, r.depth, r.alpha, r.state["n_contrib"], r.state["final_T"])
    finally:
        lib.splatraster_debug_set_fwd_team(-1)
    for mode in (-1, 2):
        for a, b in zip(outs[0], outs[mode]):
            assert torch.equal(a, b), mode
    assert int(outs[0][3].max()) > 500      # deep lists: hundreds of contributors per pixel
    # the split backward starts its later waves from the forward's segment records: those a team wrote (B: T at the boundaries,
    # C: the segments' sums) against the one-wave kernel's
    grads = {}
    try:
        for mode in (0, 2):
            lib.splatraster_debug_set_fwd_team(mode)
            r = HipRun(sc)
            grads[mode] = {n: getattr(r, n).grad.cpu().numpy() for n in ("means3D", "opacities", "colors", "scales", "rotations")}
    finally:
        lib.splatraster_debug_set_fwd_team(-1)
    for n, g0 in grads[0].items():
        assert_grad_close(n, grads[2][n], g0, rtol=1e-4, atol_scale=2e-5)


def test_team_forward_soak():
    """Random shapes through both forwards of the narrow layouts (tools/soak_team_forward.py runs hundreds): a team's waves meet at
    barriers and hand values over through LDS — a race would be a rare mismatch, not a reproducible one."""
    from splatloc_amd import _native
    lib = _native.load()
    rng = np.random.default_rng(77)
    try:
        for it in range(40):
            C, W, H = int(rng.integers(1, 5)), int(rng.integers(8, 300)), int(rng.integers(8, 200))
            sc = make_scene(int(rng.integers(1, 12000)), W, H, C, seed=int(rng.integers(1 << 30)), scale_median=float(rng.choice([0.01, 0.05, 0.2, 0.5])))
            if it % 2:
                sc.opacities = sc.opacities * float(rng.choice([0.02, 0.1, 0.5]))
            outs = {}
            for mode in (0, 2):
                lib.splatraster_debug_set_fwd_team(mode)
                r = HipRun(sc, backward=False)
                outs[mode] = (r.color, r.depth, r.alpha, r.state["n_contrib"], r.state["final_T"])
            for a, b in zip(outs[0], outs[2]):
                assert torch.equal(a, b), (it, C, W, H)
    finally:
        lib.splatraster_debug_set_fwd_team(-1)


def test_streaming_payload_stores_change_nothing():
    """Lists of >= 8 Mi instances write the per-instance payload with non-temporal stores (binning.hip); the hook forces that
    path on a small scene: same payload, images and (deterministic-sum mode) gradients, bit for bit."""
    from splatloc_amd import _native
    lib = _native.load()
    sc = make_scene(3000, 200, 120, 35, seed=77, scale_median=0.12)
    _native.set_deterministic(True)
    try:
        lib.splatraster_debug_set_payload_stream_min(0)
        a = HipRun(sc)
        lib.splatraster_debug_set_payload_stream_min(-1)
        b = HipRun(sc)
    finally:
        lib.splatraster_debug_set_payload_stream_min(-1)
        _native.set_deterministic(False)
    for k in ("irec", "imask", "ranges", "n_contrib", "final_T"):
        if k in a.state:
            assert torch.equal(a.state[k], b.state[k]), k
    assert torch.equal(a.color, b.color) and torch.equal(a.depth, b.depth) and torch.equal(a.alpha, b.alpha)
    for n in ("means3D", "means2D", "opacities", "colors", "scales", "rotations"):
        assert torch.equal(getattr(a, n).grad, getattr(b, n).grad), n


@pytest.mark.parametrize("C", [3, 4, 8, 35])
def test_results_do_not_depend_on_what_the_lds_held(C):
    """The compositing kernels pair the candidates of a staging round; an odd round's last pair has no second member and reads
    a staged feature row nobody wrote in that round.  The forward zeroes that row (composite_fwd.hip), the backward masks the
    half (composite_bwd.hip): with every compute unit's LDS pre-filled with NaNs, then with a huge finite pattern, images and
    (deterministic-sum mode) gradients are the ones of an undisturbed run, bit for bit."""
    from splatloc_amd import _native
    lib = _native.load()
    sc = make_scene(2500, 136, 104, C, seed=123 + C, scale_median=0.08)
    _native.set_deterministic(True)
    try:
        ref = HipRun(sc)
        runs = []
        for pattern in (0x7FC00000, 0x7F7FFFFF, 0xFF800000):   # NaN, FLT_MAX, -inf
            _native.check(lib.splatraster_debug_poison_lds(pattern, None), "poison_lds")
            torch.cuda.synchronize()
            runs.append(HipRun(sc))
    finally:
        _native.set_deterministic(False)
    for r in runs:
        assert torch.equal(ref.color, r.color) and torch.equal(ref.depth, r.depth) and torch.equal(ref.alpha, r.alpha)
        assert torch.isfinite(r.color).all()
        for n in ("means3D", "means2D", "opacities", "colors", "scales", "rotations"):
            assert torch.equal(getattr(ref, n).grad, getattr(r, n).grad), n
