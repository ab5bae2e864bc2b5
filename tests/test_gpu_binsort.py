"""The binned front end (csrc/binsort.hip: counting sort by (view, tile) + one LDS sort per tile) against the radix front
end (two global stable sorts) and the CPU oracle — needs an MI355X.

Both front ends must leave the SAME forward state: point list, tile ids, ranges, the per-instance payload the compositing
kernels stream, and therefore bit-identical images / n_contrib / final_T.  The per-view and window entry points share the
code; shapes are chosen so that every tier of the per-tile sort runs: the one-wave network (lists <= 256), the four-wave
network in LDS, the work list of the second launch (lists beyond four times the mean) and the global-memory fall-back
(one list beyond 16 384 entries).
"""
import numpy as np
import pytest
import torch

from splatloc_amd import _native
from splatloc_amd.synthetic import make_scene
from tests.helpers import HipRun, oracle_backward, oracle_forward
from tests.test_gpu_parity import _check_backward, _check_forward

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore_front_end():
    yield
    _native.set_front_end(-1)
    _native.check(_native.load().splatraster_debug_set_tile_sort_cap(0), "tile_sort_cap")
    _native.check(_native.load().splatraster_debug_set_sort_fork(-1), "sort_fork")


def _state_equal(a: HipRun, b: HipRun):
    for k in ("tiles_touched", "point_list", "tile_list", "ranges", "n_contrib"):
        assert torch.equal(a.state[k], b.state[k]), k
    assert torch.equal(a.state["final_T"].view(torch.int32), b.state["final_T"].view(torch.int32)), "final_T bits"
    for k in ("color", "depth", "alpha"):
        assert torch.equal(getattr(a, k).view(torch.int32), getattr(b, k).view(torch.int32)), k + " bits"
    assert torch.equal(a.radii, b.radii)


def _concentrate(sc, n, pixel, depth_lo=1.0, depth_hi=5.0, seed=7):
    """moves the first n Gaussians onto one pixel (distinct depths): one tile's list gets n entries more"""
    g = torch.Generator().manual_seed(seed)
    cam = sc.camera
    z = depth_lo + (depth_hi - depth_lo) * torch.rand(n, generator=g)
    fx, fy = cam.image_width / (2.0 * cam.tanfovx), cam.image_height / (2.0 * cam.tanfovy)
    px, py = pixel
    sc.means3D[:n, 0] = (px - (cam.image_width - 1) / 2.0) / fx * z
    sc.means3D[:n, 1] = (py - (cam.image_height - 1) / 2.0) / fy * z
    sc.means3D[:n, 2] = z
    sc.scales[:n] = 0.002
    return sc


CASES = {
    "S0": lambda: make_scene(10_000, 640, 480, 3, 0, scale_median=0.02),
    "ragged_C35": lambda: make_scene(3_000, 333, 201, 35, 22, scale_median=0.03),
    "tiny": lambda: make_scene(300, 17, 9, 4, 117, scale_median=0.08),
    "deep_lists": lambda: make_scene(20_000, 256, 256, 3, 26, scale_median=0.05),
    "screen_filling": lambda: make_scene(200, 96, 64, 3, 41, scale_median=1.5),
    "one_long_list": lambda: _concentrate(make_scene(30_000, 256, 256, 4, 51, scale_median=0.02), 9_000, (100.3, 77.6)),
    "beyond_lds": lambda: _concentrate(make_scene(24_000, 128, 128, 3, 52, scale_median=0.02), 18_000, (40.2, 50.9)),
    "more_than_one_chunk": lambda: make_scene(40_000, 320, 240, 4, 53, scale_median=0.01),
}


@pytest.mark.parametrize("name", list(CASES))
def test_front_ends_leave_the_same_state(name):
    sc = CASES[name]()
    _native.set_front_end(0)
    radix = HipRun(sc, backward=False)
    _native.set_front_end(1)
    binned = HipRun(sc, backward=True)
    _state_equal(radix, binned)
    # and the binned run on its own against the oracle, forward and backward
    f = oracle_forward(sc)
    _check_forward(binned, f, sc)
    _check_backward(binned, oracle_backward(f, sc))
    if name == "one_long_list":
        r = f["ranges"].astype(np.int64)
        lens = r[:, 1] - r[:, 0]
        assert lens.max() > 8192, (lens.max(), lens.mean())   # the work-list launch, 16 keys per thread
    if name == "beyond_lds":
        r = f["ranges"].astype(np.int64)
        assert (r[:, 1] - r[:, 0]).max() > 16384


@pytest.mark.parametrize("cap", [2048, 4096, 0])
def test_lists_between_the_two_tile_launches(cap):
    """A list of ~3 500 keys: sorted by the 256-thread instantiation of the tile launch (cap 4096), by the work-list launch
    (cap 2048), or by whichever the hint selects (0: the first frame meets the list with the narrow launch and raises the hint,
    the second frame takes the wide one) — the same state every time."""
    sc = _concentrate(make_scene(20_000, 256, 256, 4, 54, scale_median=0.02), 3_200, (130.4, 60.7))
    f = oracle_forward(sc)
    r = f["ranges"].astype(np.int64)
    assert 2048 < (r[:, 1] - r[:, 0]).max() <= 4096
    _native.set_front_end(1)
    _native.check(_native.load().splatraster_debug_set_tile_sort_cap(cap), "tile_sort_cap")
    for _ in range(2):
        run = HipRun(sc, backward=False)
        _check_forward(run, f, sc)


def test_exact_depth_ties_and_duplicates_keep_index_order():
    sc = make_scene(600, 128, 96, 3, seed=42, scale_median=0.06)
    sc.means3D[:, 2] = torch.round(sc.means3D[:, 2] * 2.0) / 2.0
    sc.means3D[300:] = sc.means3D[:300]
    _native.set_front_end(1)
    run = HipRun(sc, backward=False)
    _check_forward(run, oracle_forward(sc), sc)


def test_empty_frame_and_all_culled():
    sc = make_scene(500, 96, 64, 4, seed=61, scale_median=0.05)
    sc.means3D[:, 2] = -1.0          # everything behind the camera
    _native.set_front_end(1)
    run = HipRun(sc, backward=True)
    assert run.num_rendered == 0
    assert float(run.color.abs().max()) == 0.0
    assert int(run.state["ranges"].abs().max()) == 0


def test_window_of_views_binned_equals_radix():
    """5 views of one scene as ONE launch sequence: per-view state identical under both front ends."""
    from splatloc_amd import introspect, rasterize_window
    from tests.test_gpu_window import _views      # the window tests' camera set
    sc = make_scene(30_000, 320, 240, 4, 71, scale_median=0.02)
    dev = torch.device("cuda:0")
    settings = [rs for _, rs, _ in _views(sc, 5, dev)]
    outs = {}
    for mode in (0, 1):
        _native.set_front_end(mode)
        m3 = sc.means3D.to(dev).requires_grad_(True)
        m2 = [torch.zeros_like(m3, requires_grad=True) for _ in settings]
        res = rasterize_window(settings, m3, m2, sc.features.to(dev), sc.opacities.to(dev), sc.scales.to(dev),
                               sc.rotations.to(dev))
        fn = res[0][0].grad_fn
        saved = fn.saved_tensors
        st = introspect.window_state((saved[7], saved[8], saved[9]), 30_000, 5, 320, 240, fn.R)
        outs[mode] = (res, st)
    torch.cuda.synchronize()
    for v in range(5):
        a, b = outs[0][1][v], outs[1][1][v]
        for k in ("point_list", "tile_list", "ranges", "n_contrib"):
            assert torch.equal(a[k], b[k]), (v, k)
        assert torch.equal(a["final_T"].view(torch.int32), b["final_T"].view(torch.int32))
        for i in range(3):
            assert torch.equal(outs[0][0][v][i].view(torch.int32), outs[1][0][v][i].view(torch.int32))


def _two_stage_forward(sc, flip_front_end_between_stages=None, renders=2):
    """The two public forward stages called directly (include/splatraster.h): ONE geometry stage, then `renders` render stages
    on the same geometry buffer, each with a FRESH (garbage-filled) binning buffer.  Returns the images of every render."""
    import ctypes as C
    from splatloc_amd.rasterizer import _ptr, _stream
    lib = _native.load()
    dev = torch.device("cuda:0")
    cam = sc.camera
    P, H, W, Cn = sc.means3D.shape[0], cam.image_height, cam.image_width, sc.features.shape[1]
    t = lambda x: x.to(dev).contiguous().float()  # noqa: E731
    m3, col, opa, sca, rot = t(sc.means3D), t(sc.features), t(sc.opacities), t(sc.scales), t(sc.rotations)
    view, proj, campos, bg = t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center), t(sc.bg)
    st = _native.Settings(H, W, float(cam.tanfovx), float(cam.tanfovy), 1.0, 0, 0, Cn, int(bg.numel()), 0, 0)
    geom = torch.empty((lib.splatraster_geometry_bytes(P),), dtype=torch.uint8, device=dev)
    radii = torch.empty((P,), dtype=torch.int32, device=dev)
    stream = _stream(dev)
    R = C.c_int64(0)
    _native.check(lib.splatraster_forward_geometry(C.byref(st), P, _ptr(m3), None, _ptr(opa), _ptr(sca), _ptr(rot), None, _ptr(view),
                                                   _ptr(proj), _ptr(campos), _ptr(geom), _ptr(radii), C.byref(R), stream), "geometry")
    if flip_front_end_between_stages is not None:
        _native.set_front_end(flip_front_end_between_stages)
    outs = []
    for k in range(renders):
        nbytes = lib.splatraster_binning_bytes(P, R.value, W, H, Cn)
        binning = torch.full((nbytes,), 0xA5 if k else 0x5A, dtype=torch.uint8, device=dev)      # never zero-initialised
        img = torch.empty((lib.splatraster_image_bytes(W, H),), dtype=torch.uint8, device=dev)
        color = torch.empty((Cn, H, W), device=dev)
        depth = torch.empty((1, H, W), device=dev)
        alpha = torch.empty((1, H, W), device=dev)
        _native.check(lib.splatraster_forward_render(C.byref(st), P, R.value, _ptr(bg), _ptr(col), _ptr(geom), _ptr(binning), _ptr(img),
                                                     _ptr(color), _ptr(depth), _ptr(alpha), stream), "render")
        torch.cuda.synchronize()
        outs.append((color, depth, alpha))
    return outs, int(R.value)


def test_second_render_stage_on_the_same_geometry_starts_with_an_empty_work_list():
    """Round-5 advisor finding: the work-list counter of the binned front end lives in the GEOMETRY buffer and was zeroed only
    by the geometry stage; a second `splatraster_forward_render` on the same geometry (other features, a fresh binning buffer)
    appended behind the first render's entries and then walked uninitialised tile ids.  `one_long_list` puts a 9 000-key list
    on the work list.  Three renders with garbage-filled binning buffers: identical images, equal to the oracle's."""
    sc = CASES["one_long_list"]()
    f = oracle_forward(sc)
    _native.set_front_end(1)
    outs, R = _two_stage_forward(sc, renders=3)
    assert R == int(f["num_rendered"])
    for color, depth, alpha in outs:
        assert np.abs(color.cpu().numpy() - f["color"]).max() <= 1e-4
        assert np.abs(alpha.cpu().numpy().reshape(f["alpha"].shape) - f["alpha"]).max() <= 1e-4
    for k in (1, 2):
        for a, b in zip(outs[0], outs[k]):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))


@pytest.mark.parametrize("first,then", [(1, 0), (0, 1)])
def test_render_stage_follows_the_front_end_the_geometry_stage_chose(first, then):
    """`splatraster_debug_set_front_end` between the two stages: the render stage must keep the geometry stage's choice (the
    binned render needs the (tile, chunk) table the geometry stage built; the radix render needs the depth order)."""
    sc = CASES["deep_lists"]()
    f = oracle_forward(sc)
    _native.set_front_end(first)
    outs, _ = _two_stage_forward(sc, flip_front_end_between_stages=then, renders=1)
    color, depth, alpha = outs[0]
    assert np.abs(color.cpu().numpy() - f["color"]).max() <= 1e-4
    assert np.abs(alpha.cpu().numpy().reshape(f["alpha"].shape) - f["alpha"]).max() <= 1e-4


@pytest.mark.parametrize("fork", [0, 1])
@pytest.mark.parametrize("name", ["one_long_list", "beyond_lds", "S0"])
def test_long_list_launch_on_the_side_stream_changes_nothing(name, fork):
    """The long-list sort launch finds its lists in the scanned table itself and may run beside the tile launch on the library's
    side stream (splatraster_debug_set_sort_fork: 0 never, 1 always; default: when the scene has long lists).  Same state either
    way, three frames in a row (the side stream and its events are reused), forward and backward against the oracle."""
    sc = CASES[name]()
    f = oracle_forward(sc)
    _native.set_front_end(1)
    _native.check(_native.load().splatraster_debug_set_sort_fork(fork), "sort_fork")
    runs = [HipRun(sc, backward=(k == 2)) for k in range(3)]
    for run in runs:
        _check_forward(run, f, sc)
    _check_backward(runs[2], oracle_backward(f, sc))
    _state_equal(runs[0], runs[1])


def test_window_with_long_lists_forks_by_itself():
    """Default mode: the first frame meets a list beyond 2 048 keys and raises the hint; the following frames run the long-list
    launch on the side stream.  A window of 3 views of such a scene, five frames in a row: per-view state stays the oracle's."""
    from splatloc_amd import introspect, rasterize_window
    from tests.test_gpu_window import _views
    sc = _concentrate(make_scene(20_000, 256, 256, 4, 54, scale_median=0.02), 5_000, (130.4, 60.7))
    dev = torch.device("cuda:0")
    settings = [rs for _, rs, _ in _views(sc, 3, dev)]
    _native.set_front_end(1)
    first = None
    for _ in range(5):
        m3 = sc.means3D.to(dev).requires_grad_(True)
        m2 = [torch.zeros_like(m3, requires_grad=True) for _ in settings]
        res = rasterize_window(settings, m3, m2, sc.features.to(dev), sc.opacities.to(dev), sc.scales.to(dev), sc.rotations.to(dev))
        torch.cuda.synchronize()
        imgs = [r[0].detach().clone() for r in res]
        if first is None:
            first = imgs
        for a, b in zip(first, imgs):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))
