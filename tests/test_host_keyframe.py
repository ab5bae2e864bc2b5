"""Host-side pieces of the key-frame insertion (splatloc_amd/keyframe.py) that need no GPU: the numpy-exact median and the
replica-safe down-sampling draw (round-3 advisor findings)."""
import numpy as np
import torch

from splatloc_amd.keyframe import _keyed_draw, _np_median_f32


def test_median_is_numpys_float32_median():
    """adaptive_pointsize (gaussian_model.py:175-177): `np.median(depth)` of a float32 depth map averages the two middle
    values of an even count in FLOAT32; the float64 midpoint of torch.quantile differs in the last bit."""
    g = torch.Generator().manual_seed(3)
    differs = 0
    for n in (1, 2, 7, 8, 1000, 1001, 640 * 480):
        for _ in range(20 if n < 2000 else 2):
            a = torch.rand(n, generator=g) * 5
            m = _np_median_f32(a.reshape(1, -1))
            assert m.dtype == torch.float32
            assert np.float32(m.item()) == np.median(a.numpy()), n
            q = torch.quantile(a.double(), 0.5, interpolation="midpoint")
            differs += int(float(q) != float(m))
    assert differs > 0      # the float64 midpoint really is a different number for some even counts


def test_downsampling_draw_is_keyed_not_global():
    """`np.random.choice(n_points, n_samples)` of gaussian_model.py:159-163 — here keyed by (seed, kf_id): identical on every
    replica whatever the global RNG states are, different per key-frame."""
    a = _keyed_draw(5000, 78, seed=4, kf_id=11)
    torch.manual_seed(999)
    torch.rand(10)
    b = _keyed_draw(5000, 78, seed=4, kf_id=11)
    assert torch.equal(a, b) and a.dtype == torch.int64 and int(a.min()) >= 0 and int(a.max()) < 5000
    assert not torch.equal(a, _keyed_draw(5000, 78, seed=4, kf_id=12))
    assert not torch.equal(a, _keyed_draw(5000, 78, seed=5, kf_id=11))
    assert _keyed_draw(0, 0, 0, 0).numel() == 0
