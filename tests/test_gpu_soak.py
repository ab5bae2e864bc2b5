"""Short soak (the full one is tools/soak.py): frames of random sizes back to back with
device-side checks of the ordering contract — (tile, depth, index) order, ranges partition the
list, instance count — which a fixed-size parity test cannot provoke (sort path switches, scan
look-back under different block counts, buffer re-sizing)."""
import pytest
import torch

from splatloc_amd.synthetic import make_scene
from tests.helpers import HipRun

pytestmark = pytest.mark.gpu


def test_random_sizes_keep_the_ordering_contract():
    g = torch.Generator().manual_seed(123)
    for it in range(40):
        P = int(10 ** (torch.rand(1, generator=g).item() * 5.3))
        W = int(16 + torch.randint(0, 900, (1,), generator=g).item())
        H = int(16 + torch.randint(0, 600, (1,), generator=g).item())
        C = [1, 3, 4, 7, 35][int(torch.randint(0, 5, (1,), generator=g).item())]
        sm = 10 ** (-2.6 + 1.5 * torch.rand(1, generator=g).item())
        run = HipRun(make_scene(P, W, H, C, seed=500 + it, scale_median=sm), backward=(it % 4 == 0))
        st, R = run.state, run.num_rendered
        assert R == int(st["tiles_touched"].long().sum()), it
        assert bool(torch.isfinite(run.color).all()), it
        if not R:
            continue
        tiles, pl = st["tile_list"].long(), st["point_list"].long()
        assert bool((tiles[1:] >= tiles[:-1]).all()), it
        bits = st["rec0"][:, 2].contiguous().view(torch.int32).long()
        same = tiles[1:] == tiles[:-1]
        d0, d1 = bits[pl[:-1]], bits[pl[1:]]
        assert bool(((d1 > d0) | ((d1 == d0) & (pl[1:] > pl[:-1])))[same].all()), it
        rng = st["ranges"].long()
        assert bool((torch.bincount(tiles, minlength=rng.shape[0]) == rng[:, 1] - rng[:, 0]).all()), it


def test_split_backward_part_logic_on_random_clustered_frames():
    """Short form of tools/soak_split.py: random small frames of narrow layouts with random clusters (lists of a few hundred to several
    thousand entries: 4 / 8 / 16 parts, lengths around the thresholds), all three front-end modes: the split backward against the
    one-wave-per-quadrant backward — images bit-identical, gradients within the per-row bars."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_split.py"), "15", "3"], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "soak_split ok" in r.stdout, (r.stdout[-800:], r.stderr[-2000:])
