"""Generates tests/golden/pose_helpers.npz from the reference's OWN utils/optimization_utils.py, imported in THIS container.

The module's first line imports four functions from pytorch3d (not installed): `pytorch3d.transforms` is stubbed with
placeholders, so only the two functions whose bodies do not touch pytorch3d are recorded:

    axis_angle_to_matrix(data)            optimization_utils.py:5-22     (Rodrigues; NaN at the zero vector, its own TODO)
    at_to_transform_matrix(rot, trans)    optimization_utils.py:31-42

on seeded inputs: random axis-angle vectors with angles up to ~3 rad, tiny ones (1e-4 rad), and the zero vector (recorded
as the NaN the reference produces; the build returns the identity there and says so).  The quaternion / 6-D helpers of the
same file are thin wrappers around pytorch3d and are pinned against scipy in tests/test_host_pose.py instead.
Only the fixture (data) is committed; nothing of the reference travels.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    def _absent(*a, **k):
        raise RuntimeError("pytorch3d is not installed: this helper is not recorded")
    p3d = types.ModuleType("pytorch3d")
    tr = types.ModuleType("pytorch3d.transforms")
    for n in ("matrix_to_quaternion", "quaternion_to_matrix", "rotation_6d_to_matrix", "quaternion_to_axis_angle"):
        setattr(tr, n, _absent)
    p3d.transforms = tr
    sys.modules["pytorch3d"], sys.modules["pytorch3d.transforms"] = p3d, tr
    sys.path.insert(0, "/root/reference")
    from utils import optimization_utils as ou
    g = torch.Generator().manual_seed(4242)
    w = torch.randn(64, 3, generator=g) * torch.rand(64, 1, generator=g) * 1.8
    w[60] = torch.tensor([1e-4, -2e-4, 5e-5])
    w[61] = torch.tensor([0.0, 0.0, 3.0])
    w[62] = torch.tensor([1e-7, 0.0, 0.0])
    w[63] = 0.0
    t = torch.randn(64, 3, generator=g)
    out = {"w": w.numpy(), "t": t.numpy(), "R": ou.axis_angle_to_matrix(w).numpy(), "T": ou.at_to_transform_matrix(w, t).numpy()}
    assert np.isnan(out["R"][63]).all()          # the reference's behaviour at the zero vector
    # a batched [2, 5, 3] input: the function is written for arbitrary batch dimensions
    wb = w[:10].reshape(2, 5, 3)
    out["w_batched"], out["R_batched"] = wb.numpy(), ou.axis_angle_to_matrix(wb).numpy()
    np.savez_compressed(os.path.join(HERE, "pose_helpers.npz"), **out)
    print("wrote pose_helpers.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
