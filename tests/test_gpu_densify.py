"""Device-side densification statistics against the fixture recorded from the reference."""
import numpy as np
import pytest
import torch

from tests.test_oracle_stats import GOLD

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_densification_stats_match_reference():
    from splatloc_amd.densify import add_densification_stats
    d = np.load(GOLD)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
    a, n, m = t(d["accum0"]), t(d["denom0"]), t(d["max_radii0"])
    for v in range(3):
        add_densification_stats(t(d[f"grad{v}"]), t(d[f"radii{v}"]), a, n, m)
        np.testing.assert_allclose(a.cpu().numpy(), d[f"accum{v + 1}"], rtol=1e-6, atol=1e-9)
        assert np.array_equal(n.cpu().numpy(), d[f"denom{v + 1}"])
        assert np.array_equal(m.cpu().numpy(), d[f"max_radii{v + 1}"])
    with pytest.raises(RuntimeError):
        add_densification_stats(t(d["grad0"]), t(d["radii0"]).float(), a, n, m)
    with pytest.raises(RuntimeError):
        add_densification_stats(t(d["grad0"]).cpu(), t(d["radii0"]), a, n, m)
