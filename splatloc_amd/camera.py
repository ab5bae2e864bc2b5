"""Camera matrices in the convention the rasterizer consumes.

Own restatement of what SplatLoc's Python produces for the rasterizer settings
(utils/camera_utils.py:129-139, gaussian_splatting/utils/graphics_utils.py:33-46,72-93);
pinned against the reference by tests/golden/camera.npz.  Row-vector convention:
p_view = [p 1] @ world_view_transform, p_clip = [p 1] @ full_proj_transform.
"""
from __future__ import annotations

import math

import torch


def projection_matrix(znear: float, zfar: float, fx: float, fy: float, cx: float, cy: float, W: int,
                      H: int) -> torch.Tensor:
    """OpenCV-intrinsics perspective matrix (column-vector form, like getProjectionMatrix2)."""
    P = torch.zeros(4, 4, dtype=torch.float32)
    P[0, 0] = 2.0 * fx / W
    P[1, 1] = 2.0 * fy / H
    P[0, 2] = (2.0 * cx - W) / W
    P[1, 2] = (2.0 * cy - H) / H
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def world_view_transform(R: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """[4,4] row-vector world->view matrix from a W2C rotation R and translation t."""
    Rt = torch.eye(4, dtype=torch.float32)
    Rt[:3, :3] = R
    Rt[:3, 3] = t
    return Rt.transpose(0, 1).contiguous()


class PinholeCamera:
    """The three tensors + two scalars the rasterizer settings need."""

    def __init__(self, W: int, H: int, fx: float, fy: float, cx: float, cy: float,
                 R: torch.Tensor | None = None, t: torch.Tensor | None = None, znear: float = 0.01,
                 zfar: float = 100.0):
        self.image_width, self.image_height = W, H
        self.fx, self.fy, self.cx, self.cy = fx, fy, cx, cy
        self.FoVx = 2.0 * math.atan(W / (2.0 * fx))
        self.FoVy = 2.0 * math.atan(H / (2.0 * fy))
        self.tanfovx = math.tan(self.FoVx * 0.5)
        self.tanfovy = math.tan(self.FoVy * 0.5)
        R = torch.eye(3) if R is None else R.float()
        t = torch.zeros(3) if t is None else t.float()
        self.world_view_transform = world_view_transform(R, t)
        self.projection_matrix = projection_matrix(znear, zfar, fx, fy, cx, cy, W, H).transpose(0, 1)
        self.full_proj_transform = self.world_view_transform @ self.projection_matrix
        self.camera_center = torch.linalg.inv(self.world_view_transform)[3, :3].contiguous()

    def to(self, device):
        for k in ("world_view_transform", "projection_matrix", "full_proj_transform", "camera_center"):
            setattr(self, k, getattr(self, k).to(device))
        return self
