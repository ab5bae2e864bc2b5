"""oracle/stats.py against the fixture recorded from the reference's GaussianModel."""
import os

import numpy as np

from oracle import stats

GOLD = os.path.join(os.path.dirname(__file__), "golden", "densify_stats.npz")


def test_densification_stats_match_reference():
    d = np.load(GOLD)
    a, n, m = d["accum0"], d["denom0"], d["max_radii0"]
    for v in range(3):
        a, n, m = stats.densification_stats(d[f"grad{v}"], d[f"radii{v}"], a, n, m)
        np.testing.assert_allclose(a, d[f"accum{v + 1}"], rtol=1e-6, atol=1e-9)
        assert np.array_equal(n, d[f"denom{v + 1}"]) and np.array_equal(m, d[f"max_radii{v + 1}"])
