"""Soak run (GPU): many frames of random sizes back to back, checking on the device that the
instance list is sorted by (tile, depth, index), that the ranges partition it, that the instance
count matches, and that nothing is NaN.  Catches rare ordering / look-back / sizing problems that
a fixed-size test would miss.  usage: python tools/soak.py [frames] [seed]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from splatloc_amd import introspect  # noqa: E402
from splatloc_amd.synthetic import make_scene  # noqa: E402
from tests.helpers import HipRun  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.time()
worst = 0
for it in range(frames):
    P = int(10 ** (torch.rand(1, generator=g).item() * 5.5))            # 1 .. 316k
    W = int(16 + torch.randint(0, 1200, (1,), generator=g).item())
    H = int(16 + torch.randint(0, 700, (1,), generator=g).item())
    C = [1, 3, 4, 7, 35][int(torch.randint(0, 5, (1,), generator=g).item())]
    sm = 10 ** (-2.6 + 1.6 * torch.rand(1, generator=g).item())
    sc = make_scene(P, W, H, C, seed=1000 + it, scale_median=sm)
    run = HipRun(sc, backward=(it % 3 == 0))
    st = run.state
    R = run.num_rendered
    assert R == int(st["tiles_touched"].long().sum()), (it, "R")
    if R:
        tiles = st["tile_list"].long()
        pl = st["point_list"].long()
        assert bool((tiles[1:] >= tiles[:-1]).all()), (it, "tile order")
        depth_bits = st["rec0"][:, 2].contiguous().view(torch.int32).long()
        same = tiles[1:] == tiles[:-1]
        d0, d1 = depth_bits[pl[:-1]], depth_bits[pl[1:]]
        assert bool(((d1 > d0) | ((d1 == d0) & (pl[1:] > pl[:-1])))[same].all()), (it, "(depth, index) order")
        rng = st["ranges"].long()
        assert int((rng[:, 1] - rng[:, 0]).sum()) == R, (it, "ranges")
        cnt = torch.bincount(tiles, minlength=rng.shape[0])
        assert bool((cnt == rng[:, 1] - rng[:, 0]).all()), (it, "range sizes")
    assert bool(torch.isfinite(run.color).all()), (it, "nan")
    if it % 3 == 0:
        for t in (run.means3D, run.colors, run.opacities, run.scales, run.rotations):
            assert bool(torch.isfinite(t.grad).all()), (it, "nan grad")
    worst = max(worst, R)
print(f"soak ok: {frames} frames, largest R = {worst}, {time.time() - t0:.1f} s")

# one frame with more than 67 M instances: the large sort's per-pass scan takes its 3-kernel path
sc = make_scene(60_000, 1280, 704, 3, seed=77, scale_median=0.6)
run = HipRun(sc, backward=False)
st = run.state
R = run.num_rendered
tiles = st["tile_list"].long()
assert R == int(st["tiles_touched"].long().sum()) and bool((tiles[1:] >= tiles[:-1]).all())
rng = st["ranges"].long()
assert bool((torch.bincount(tiles, minlength=rng.shape[0]) == rng[:, 1] - rng[:, 0]).all())
pl = st["point_list"].long()
depth_bits = st["rec0"][:, 2].contiguous().view(torch.int32).long()
same = tiles[1:] == tiles[:-1]
assert bool(((depth_bits[pl[1:]] > depth_bits[pl[:-1]]) | ((depth_bits[pl[1:]] == depth_bits[pl[:-1]]) & (pl[1:] > pl[:-1])))[same].all())
print(f"huge-R frame ok: R = {R}")
