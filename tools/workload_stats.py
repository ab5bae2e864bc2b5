#!/usr/bin/env python3
"""Sparsity statistics of a workload's compositing (guides kernel design; CPU oracle only).
For a sample of tiles: list length, processed prefix, fraction of (8x8 quadrant, Gaussian)
steps with at least one contributing pixel, pixel hit density."""
import sys
import numpy as np
sys.path.insert(0, ".")
from splatloc_amd.synthetic import make_workload
from tests.helpers import oracle_forward

name = sys.argv[1] if len(sys.argv) > 1 else "S2"
sc = make_workload(name)
f = oracle_forward(sc)
W, H = sc.camera.image_width, sc.camera.image_height
gx = (W + 15) // 16
rng = np.random.default_rng(0)
tiles = rng.choice(len(f["ranges"]), 300, replace=False)
tot_len = tot_proc = 0
steps = {8: [0, 0], 4: [0, 0]}   # quadrant edge -> [processed sub-steps, sub-steps with any hit]
hits = 0
for t in tiles:
    s, e = f["ranges"][t]
    ty, tx = divmod(t, gx)
    ys, xs = np.mgrid[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16]
    inside = (ys < H) & (xs < W)
    nc = np.where(inside, f["n_contrib"][np.minimum(ys, H - 1), np.minimum(xs, W - 1)], 0)
    last = nc.max()
    n = min(e - s, last)
    tot_len += e - s
    tot_proc += n
    if n == 0:
        continue
    g = f["point_list"][s:s + n]
    xy = f["xy"][g]; con = f["conic_opacity"][g]
    dx = xy[:, 0, None, None] - xs[None]; dy = xy[:, 1, None, None] - ys[None]
    power = -0.5 * (con[:, 0, None, None] * dx * dx + con[:, 2, None, None] * dy * dy) - con[:, 1, None, None] * dx * dy
    alpha = np.minimum(0.99, con[:, 3, None, None] * np.exp(np.minimum(power, 0)))
    idx = np.arange(1, n + 1)[:, None, None]
    hit = (power <= 0) & (alpha >= 1 / 255) & (idx <= nc[None]) & inside[None]
    hits += hit.sum()
    for q in (8, 4):
        hq = hit.reshape(n, 16 // q, q, 16 // q, q).any(axis=(2, 4))           # [n, 16/q, 16/q]
        active = (idx <= nc.reshape(16 // q, q, 16 // q, q).max(axis=(1, 3))[None])
        steps[q][0] += active.sum()
        steps[q][1] += hq.sum()
print(f"{name}: R={f['num_rendered']}  sampled tiles={len(tiles)}")
print(f" mean list length {tot_len / len(tiles):.1f}, processed prefix (<= tile's deepest contributor) {tot_proc / len(tiles):.1f}"
      f" = {100 * tot_proc / tot_len:.1f}% of instances")
for q in (8, 4):
    a, h = steps[q]
    print(f" {q}x{q} sub-tile steps: processed {a / len(tiles):.0f}/tile, with >=1 hit {h / len(tiles):.0f}/tile ({100 * h / max(a, 1):.1f}%)")
print(f" pixel hits per tile {hits / len(tiles):.0f}; hits per processed instance {hits / max(tot_proc, 1):.1f} of 256"
      f"; lanes active in hit 8x8 steps {hits / max(steps[8][1], 1):.1f} of 64")

# ---- reach-mask candidates (same test as composite_common.h::quadrant_reach_mask) ----
def reach_mask(xy, con, tx0, ty0):
    A, B, C, o = con[:, 0], con[:, 1], con[:, 2], con[:, 3]
    with np.errstate(all="ignore"):
        lim0 = 2 * np.log(255 * o)
        lim = lim0 + 0.02 + 1e-4 * np.abs(lim0)
        nBrA, nBrC = -B / A, -B / C
        out = np.zeros((len(A), 4), bool)
        for q in range(4):
            bx0, by0 = tx0 + (q & 1) * 8, ty0 + (q >> 1) * 8
            u0, u1 = xy[:, 0] - (bx0 + 7), xy[:, 0] - bx0
            v0, v1 = xy[:, 1] - (by0 + 7), xy[:, 1] - by0
            inside = (u0 <= 0) & (u1 >= 0) & (v0 <= 0) & (v1 >= 0)
            qmin = np.full(len(A), 3e38)
            for ue in (u0, u1):
                vs = np.minimum(v1, np.maximum(v0, nBrC * ue))
                qmin = np.minimum(qmin, A * ue * ue + 2 * B * ue * vs + C * vs * vs)
            for ve in (v0, v1):
                us = np.minimum(u1, np.maximum(u0, nBrA * ve))
                qmin = np.minimum(qmin, A * us * us + 2 * B * us * ve + C * ve * ve)
            qmin = np.where(inside, 0, qmin)
            out[:, q] = (qmin <= lim) & (lim > 0)
    return out

cand_all = cand_proc = anyq = fwd_steps = 0
for t in tiles:
    s, e = f["ranges"][t]
    ty, tx = divmod(t, gx)
    g = f["point_list"][s:e]
    m = reach_mask(f["xy"][g], f["conic_opacity"][g], tx * 16, ty * 16)
    cand_all += m.sum()
    anyq += m.any(1).sum()
    ys, xs = np.mgrid[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16]
    inside = (ys < H) & (xs < W)
    nc = np.where(inside, f["n_contrib"][np.minimum(ys, H - 1), np.minimum(xs, W - 1)], 0)
    fT = np.where(inside, f["final_T"][np.minimum(ys, H - 1), np.minimum(xs, W - 1)], 0)
    for q in range(4):
        qs = (slice((q >> 1) * 8, (q >> 1) * 8 + 8), slice((q & 1) * 8, (q & 1) * 8 + 8))
        wl = nc[qs].max()
        cand_proc += m[:wl, q].sum()
print(f" reach-mask candidates per tile: all list {cand_all / len(tiles):.0f} quadrant-steps ({anyq / len(tiles):.0f} Gaussians reach >=1 quadrant of {tot_len / len(tiles):.0f});"
      f" within the backward prefix {cand_proc / len(tiles):.0f}  (hit steps {steps[8][1] / len(tiles):.0f})")
