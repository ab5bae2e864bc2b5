"""splatloc_amd.pose — the mirror of the reference's utils/optimization_utils.py — on the CPU: against the fixture recorded
from the reference's own module (tests/golden/make_golden_pose_helpers.py: the two functions that do not need pytorch3d) and,
for the quaternion / 6-D conversions pytorch3d provides there, against scipy.spatial.transform.Rotation."""
import os

import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation

from splatloc_amd import pose

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "pose_helpers.npz"))


def test_axis_angle_and_transform_match_the_reference_module():
    w, t = torch.from_numpy(GOLD["w"]), torch.from_numpy(GOLD["t"])
    R = pose.axis_angle_to_matrix(w).numpy()
    ok = ~np.isnan(GOLD["R"]).any(axis=(1, 2))
    assert ok.sum() == 63                                      # the reference is NaN at the zero vector only
    np.testing.assert_allclose(R[ok], GOLD["R"][ok], rtol=0, atol=1e-6)      # float32, a differently associated but equal formula: a few ulp
    np.testing.assert_array_equal(R[~ok], np.eye(3, dtype=np.float32)[None])      # here: the identity (module header)
    T = pose.at_to_transform_matrix(w, t).numpy()
    np.testing.assert_allclose(T[ok], GOLD["T"][ok], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(T[:, 3], np.tile(np.array([0, 0, 0, 1], dtype=np.float32), (64, 1)))
    np.testing.assert_allclose(pose.axis_angle_to_matrix(torch.from_numpy(GOLD["w_batched"])).numpy(), GOLD["R_batched"], atol=1e-6)


def test_gradient_at_the_zero_rotation_is_finite_and_first_order():
    w = torch.zeros(1, 3, dtype=torch.float64, requires_grad=True)
    R = pose.axis_angle_to_matrix(w)
    (R[0, 2, 1] - R[0, 1, 2]).backward()                       # = 2 w_x to first order
    np.testing.assert_allclose(w.grad.numpy(), [[2.0, 0.0, 0.0]], atol=1e-12)


def test_quaternion_and_axis_angle_conversions_against_scipy():
    rng = np.random.default_rng(7)
    q = rng.normal(size=(200, 4))
    q[:3] = [[1, 0, 0, 0], [0, 1, 0, 0], [1e-9, 0, 0, 1]]
    qs = q / np.linalg.norm(q, axis=1, keepdims=True)
    ref = Rotation.from_quat(qs[:, [1, 2, 3, 0]])              # scipy: scalar LAST
    R = pose.quaternion_to_matrix(torch.from_numpy(q)).numpy()       # any non-zero norm
    np.testing.assert_allclose(R, ref.as_matrix(), atol=1e-12)
    back = pose.matrix_to_quaternion(torch.from_numpy(ref.as_matrix())).numpy()
    sign = np.where(qs[:, :1] < 0, -1.0, 1.0)
    flip = np.abs(qs[:, 0]) < 1e-8                             # real part ~ 0: q and -q both have a "non-negative" real part
    np.testing.assert_allclose(back[~flip], (qs * sign)[~flip], atol=1e-9)
    np.testing.assert_allclose(np.abs((back[flip] * qs[flip]).sum(1)), 1.0, atol=1e-9)
    aa = pose.matrix_to_axis_angle(torch.from_numpy(ref.as_matrix())).numpy()
    np.testing.assert_allclose(Rotation.from_rotvec(aa).as_matrix(), ref.as_matrix(), atol=1e-9)
    tiny = Rotation.from_rotvec(rng.normal(size=(50, 3)) * 1e-8)
    np.testing.assert_allclose(pose.matrix_to_axis_angle(torch.from_numpy(tiny.as_matrix())).numpy(), tiny.as_rotvec(), atol=1e-14)
    T = pose.qt_to_transform_matrix(torch.from_numpy(q), torch.from_numpy(rng.normal(size=(200, 3))))
    np.testing.assert_allclose(T[:, :3, :3].numpy(), ref.as_matrix(), atol=1e-12)


def test_six_d_rotations_are_rotations_and_the_transform_is_returned():
    g = torch.Generator().manual_seed(3)
    d6 = torch.randn(100, 6, generator=g, dtype=torch.float64)
    R = pose.rotation_6d_to_matrix(d6)
    eye = torch.eye(3, dtype=torch.float64).expand(100, 3, 3)
    np.testing.assert_allclose((R @ R.transpose(1, 2)).numpy(), eye.numpy(), atol=1e-12)
    np.testing.assert_allclose(torch.linalg.det(R).numpy(), 1.0, atol=1e-12)
    np.testing.assert_allclose(R[:, 0].numpy(), torch.nn.functional.normalize(d6[:, :3], dim=1).numpy(), atol=1e-15)   # ROW 0 = b1
    t = torch.randn(100, 3, generator=g, dtype=torch.float64)
    T = pose.six_t_to_transform_matrix(d6, t)                  # (the reference's version ends in a bare `return`)
    assert T is not None and tuple(T.shape) == (100, 4, 4)
    np.testing.assert_array_equal(T[:, :3, 3].numpy(), t.numpy())


def test_camera_tensors_are_the_cameras():
    from splatloc_amd.camera import PinholeCamera
    R = torch.from_numpy(Rotation.from_rotvec([0.2, -0.4, 0.1]).as_matrix()).float()
    t = torch.tensor([0.3, -0.1, 0.8])
    cam = PinholeCamera(640, 480, 320.0, 320.0, 319.5, 239.5, R, t)
    W2C = torch.eye(4)
    W2C[:3, :3], W2C[:3, 3] = R, t
    view, proj, campos = pose.camera_tensors(W2C, cam.projection_matrix)
    np.testing.assert_allclose(view.numpy(), cam.world_view_transform.numpy(), atol=0)
    np.testing.assert_allclose(proj.numpy(), cam.full_proj_transform.numpy(), atol=1e-6)
    np.testing.assert_allclose(campos.numpy(), cam.camera_center.numpy(), atol=1e-6)
