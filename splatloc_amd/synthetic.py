"""Seeded synthetic scenes S0/S1/S2 of BASELINE.md §2 / SURVEY.md §8d.

Camera: identity pose, fx = fy = W/2 (tanfovx = 1), principal point ((W-1)/2, (H-1)/2),
znear 0.01, zfar 100.  Gaussians: xyz uniform in the frustum slab z in [0.5, 6] m with
x, y in +-1.1 z tanfov; per-axis log-scale ~ N(log SCALE_MEDIAN, 0.5^2); quaternion =
normalised N(0,1)^4; opacity = sigmoid(N(0, 1.5^2)); features ~ U(0,1);
dL/dout ~ U(-1,1)/(H W).  Everything is generated on the CPU with
torch.Generator().manual_seed(seed) so every machine sees identical inputs.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

from .camera import PinholeCamera

# "the scale median is tuned once so that tile instances R ~= 8 P at 1080p and
# then frozen".  Tuned with the oracle: S2 (500k, 1920x1080) gives R = 4.0 M at 0.00627.
SCALE_MEDIAN = 0.02
SCALE_SIGMA = 0.5


@dataclass
class Scene:
    camera: PinholeCamera
    means3D: torch.Tensor     # [P,3]
    scales: torch.Tensor      # [P,3] (activated: positive)
    rotations: torch.Tensor   # [P,4] unit quaternions (w,x,y,z)
    opacities: torch.Tensor   # [P,1] in (0,1)
    features: torch.Tensor    # [P,C]
    bg: torch.Tensor          # [min(C,3)]
    dL_dcolor: torch.Tensor   # [C,H,W]
    dL_ddepth: torch.Tensor   # [1,H,W]
    dL_dalpha: torch.Tensor   # [1,H,W]

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        self.camera.to(device)
        return self


def make_scene(P: int, W: int, H: int, C: int, seed: int, scale_median: float = SCALE_MEDIAN) -> Scene:
    g = torch.Generator().manual_seed(seed)
    fx = fy = W / 2.0
    cam = PinholeCamera(W, H, fx, fy, (W - 1) / 2.0, (H - 1) / 2.0)
    z = 0.5 + 5.5 * torch.rand(P, generator=g)
    x = (2.0 * torch.rand(P, generator=g) - 1.0) * 1.1 * z * cam.tanfovx
    y = (2.0 * torch.rand(P, generator=g) - 1.0) * 1.1 * z * cam.tanfovy
    means3D = torch.stack([x, y, z], dim=1).contiguous()
    scales = torch.exp(math.log(scale_median) + SCALE_SIGMA * torch.randn(P, 3, generator=g))
    q = torch.randn(P, 4, generator=g)
    rotations = q / q.norm(dim=1, keepdim=True)
    opacities = torch.sigmoid(1.5 * torch.randn(P, 1, generator=g))
    features = torch.rand(P, C, generator=g)
    bg = torch.zeros(min(C, 3))
    dL_dcolor = (2.0 * torch.rand(C, H, W, generator=g) - 1.0) / (H * W)
    dL_ddepth = (2.0 * torch.rand(1, H, W, generator=g) - 1.0) / (H * W)
    dL_dalpha = (2.0 * torch.rand(1, H, W, generator=g) - 1.0) / (H * W)
    return Scene(cam, means3D, scales, rotations, opacities, features, bg, dL_dcolor, dL_ddepth, dL_dalpha)


# the named workloads of BASELINE.md
WORKLOADS = {
    "S0": dict(P=10_000, W=640, H=480, C=3, seed=0, scale_median=0.02),          # R/P = 8.8
    "S1": dict(P=300_000, W=1200, H=680, C=3, seed=1, scale_median=0.010),
    "S2": dict(P=500_000, W=1920, H=1080, C=35, seed=2, scale_median=0.00627),   # R = 4.0 M (R/P = 8)
    "S2-ref-layout": dict(P=500_000, W=640, H=480, C=4, seed=2, scale_median=0.00627),
    "S1-640": dict(P=300_000, W=640, H=480, C=3, seed=1, scale_median=0.010),                 # S1 at the reference resolution
    "S2-640": dict(P=500_000, W=640, H=480, C=35, seed=2, scale_median=0.00627),              # S2 channels at the reference resolution
}


def make_workload(name: str) -> Scene:
    return make_scene(**WORKLOADS[name])
