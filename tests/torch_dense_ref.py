"""Independent dense restatement of the rasterizer spec in PyTorch (fp64, autograd).

TEST INFRASTRUCTURE.  Written against SURVEY.md §8a as straight tensor algebra — every
pixel against every Gaussian, no tiles lists, no hand-written derivative — so that the
hand-written backward of oracle/splat_oracle.c (and through it the HIP kernels) can be
checked against torch.autograd.  Only usable at toy sizes (H*W*P elements in fp64).

Lineage conventions reproduced on purpose (they define the reference gradient):
  * alpha = min(0.99, o G) is differentiated as o G (clamp is straight-through);
  * the 1.3 tanfov clamp of the view-space mean zeroes the clamped component's gradient;
  * the quaternion is used as passed (not re-normalised);
  * means2D gradient is reported in NDC units (d/d ndc, i.e. pixel gradient x 0.5 W, 0.5 H).
"""
from __future__ import annotations

import math

import torch

TILE = 16
C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792,
      0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
      -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def eval_sh_rgb(deg, sh, dirs):
    """sh [P,M,3], dirs [P,3] unit -> rgb [P,3] (+0.5, clamped at 0)."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = C0 * sh[:, 0]
    if deg > 0:
        res = res - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6]
               + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        res = (res + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10]
               + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
               + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + C3[5] * z * (xx - yy) * sh[:, 14]
               + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return torch.clamp_min(res + 0.5, 0.0)


def quat_to_rot(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.view(-1, 3, 3)


def render_dense(H, W, tanfovx, tanfovy, bg, means3D, opacities, viewmatrix, projmatrix, campos=None,
                 colors_precomp=None, shs=None, sh_degree=0, scales=None, rotations=None,
                 cov3D_precomp=None, scale_modifier=1.0, means2D_probe=None, rows=None):
    """Returns (color [C,H,W], depth [1,H,W], alpha [1,H,W], radii [P]).

    rows = (y0, y1): only the image rows [y0, y1) of the H x W frame are composited (outputs [C, y1-y0, W], ...): the
    bounded sample bench.py's CPU leg times at S0, where the full frame is 3e9 (pixel, Gaussian) pairs.

    means2D_probe: optional [P,2] zero tensor added to the NDC centre, so that its .grad is
    the dL/dmeans2D the extension reports.
    """
    dt = means3D.dtype
    dev = means3D.device      # fp64 on the GPU as well (tests/test_gpu_dense_ref.py): every tensor follows means3D
    P = means3D.shape[0]
    V, PM = viewmatrix.to(dt), projmatrix.to(dt)
    ones = torch.ones(P, 1, dtype=dt, device=dev)
    hom = torch.cat([means3D, ones], dim=1)
    pv = hom @ V          # [P,4] view space (row-vector convention)
    ph = hom @ PM
    tz = pv[:, 2]
    in_front = tz > 0.2
    pw = 1.0 / (ph[:, 3] + 1e-7)
    ndc = ph[:, :2] * pw[:, None]
    if means2D_probe is not None:
        ndc = ndc + means2D_probe
    if cov3D_precomp is not None:
        c = cov3D_precomp
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]],
                            dim=1).view(P, 3, 3)
    else:
        L = quat_to_rot(rotations) * (scale_modifier * scales)[:, None, :]
        Sigma = L @ L.transpose(1, 2)
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txtz, tytz = pv[:, 0] / tz, pv[:, 1] / tz
    cx_ = (txtz < -limx) | (txtz > limx)
    cy_ = (tytz < -limy) | (tytz > limy)
    tx = torch.where(cx_, (txtz.clamp(-limx, limx) * tz).detach(), pv[:, 0])
    ty = torch.where(cy_, (tytz.clamp(-limy, limy) * tz).detach(), pv[:, 1])
    # (the lineage treats a clamped t.x / t.y as a constant but keeps t.z live in J)
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz), zero, fy / tz, -(fy * ty) / (tz * tz)], dim=1).view(P, 2, 3)
    Wv = V[:3, :3].transpose(0, 1)  # Wv[r][c] = V[c][r]
    A = J @ Wv
    cov2 = A @ Sigma @ A.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    ok = in_front & (det != 0)
    det_s = torch.where(det != 0, det, torch.ones_like(det))
    con_a, con_b, con_c = c / det_s, -b / det_s, a / det_s
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    pix = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], dim=1)
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    pd = pix.detach()
    rminx = torch.trunc((pd[:, 0] - radius) / TILE).clamp(0, gx)
    rminy = torch.trunc((pd[:, 1] - radius) / TILE).clamp(0, gy)
    rmaxx = torch.trunc((pd[:, 0] + radius + TILE - 1) / TILE).clamp(0, gx)
    rmaxy = torch.trunc((pd[:, 1] + radius + TILE - 1) / TILE).clamp(0, gy)
    ok = ok & ((rmaxx - rminx) * (rmaxy - rminy) > 0)
    radii = torch.where(ok, radius, torch.zeros_like(radius)).to(torch.int32)

    if shs is not None:
        d = means3D - campos.to(device=dev, dtype=dt)[None]
        feat = eval_sh_rgb(sh_degree, shs, d / d.norm(dim=1, keepdim=True))
    else:
        feat = colors_precomp
    Cn = feat.shape[1]

    # global (depth, index) order == per-tile order of the spec
    keyd = torch.where(ok, tz.detach().to(torch.float32).double(), torch.full_like(tz, float("inf")).double())
    order = torch.argsort(keyd, stable=True)
    order = order[ok[order]]
    y0, y1 = (0, H) if rows is None else (int(rows[0]), int(rows[1]))
    ys, xs = torch.meshgrid(torch.arange(y0, y1, dtype=dt, device=dev), torch.arange(W, dtype=dt, device=dev), indexing="ij")
    pxf, pyf = xs.reshape(-1), ys.reshape(-1)                 # [N]
    tpx, tpy = torch.div(pxf, TILE, rounding_mode="floor"), torch.div(pyf, TILE, rounding_mode="floor")
    o = order
    in_rect = ((tpx[:, None] >= rminx[o][None]) & (tpx[:, None] < rmaxx[o][None])
               & (tpy[:, None] >= rminy[o][None]) & (tpy[:, None] < rmaxy[o][None]))     # [N,G]
    dx = pix[o, 0][None] - pxf[:, None]
    dy = pix[o, 1][None] - pyf[:, None]
    power = -0.5 * (con_a[o][None] * dx * dx + con_c[o][None] * dy * dy) - con_b[o][None] * dx * dy
    Gv = torch.exp(torch.clamp_max(power, 0.0))
    raw = opacities.reshape(-1)[o][None] * Gv
    alpha = raw + (torch.clamp_max(raw, 0.99) - raw).detach()   # straight-through clamp
    live = in_rect & (power <= 0) & (alpha.detach() >= 1.0 / 255.0)
    alpha = torch.where(live, alpha, torch.zeros_like(alpha))
    one_m = 1.0 - alpha
    Tincl = torch.cumprod(one_m, dim=1)                          # T after each Gaussian
    stop = (live & (Tincl.detach() < 1e-4)).to(torch.int8)
    stopped = torch.cummax(stop, dim=1).values.bool()            # at or after the first stop
    alpha = torch.where(stopped, torch.zeros_like(alpha), alpha)
    one_m = 1.0 - alpha
    Tincl = torch.cumprod(one_m, dim=1)
    Texcl = torch.cat([torch.ones(Tincl.shape[0], 1, dtype=dt, device=dev), Tincl[:, :-1]], dim=1)
    w = alpha * Texcl                                            # [N,G]
    Tfin = Tincl[:, -1] if Tincl.shape[1] else torch.ones((y1 - y0) * W, dtype=dt, device=dev)
    bgf = torch.zeros(Cn, dtype=dt, device=dev)
    nb = min(Cn, bg.numel())
    bgf[:nb] = bg.to(device=dev, dtype=dt)[:nb]
    color = (w @ feat[o]) + Tfin[:, None] * bgf[None]
    depth = w @ tz[o]
    alpha_img = 1.0 - Tfin
    Hb = y1 - y0
    return (color.t().reshape(Cn, Hb, W), depth.reshape(1, Hb, W), alpha_img.reshape(1, Hb, W), radii)
