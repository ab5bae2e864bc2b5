"""CPU oracle of the per-view densification statistics (first part) — numpy
restatement of gaussian_model.py:677-679 (add_densification_stats) and of the max_radii2D update
of the mapping loop (train_gaussians.py:238-245).

TEST INFRASTRUCTURE ONLY.  Parity status: PINNED by tests/golden/densify_stats.npz (recorded from
the reference's own GaussianModel, tests/golden/make_golden_densify_stats.py).
"""
import numpy as np


def densification_stats(viewspace_grad, radii, accum, denom, max_radii2D):
    """Returns the updated (accum [P,1], denom [P,1], max_radii2D [P]) for one view."""
    g = np.asarray(viewspace_grad, np.float32)
    radii = np.asarray(radii)
    vis = radii > 0
    accum, denom, mr = np.array(accum, np.float32), np.array(denom, np.float32), np.array(max_radii2D, np.float32)
    norm = np.sqrt(g[:, 0] * g[:, 0] + g[:, 1] * g[:, 1]).astype(np.float32)
    accum[vis, 0] += norm[vis]
    denom[vis, 0] += 1.0
    mr[vis] = np.maximum(mr[vis], radii[vis].astype(np.float32))
    return accum, denom, mr
