#!/usr/bin/env python3
"""Soak of the split backward's part logic (GPU): random small frames of narrow layouts with random clusters (tile lists of a few hundred
to several thousand entries: 4 / 8 / 16 parts, list lengths around the thresholds), the split backward against the one-wave-per-quadrant
backward on the same forward — images bit-identical, gradients within the per-row bars of tests/test_gpu_edge_cases.py.
usage: python tools/soak_split.py [scenes=40] [seed=0]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from splatloc_amd import _native  # noqa: E402
from tests.helpers import HipRun, assert_grad_close, assert_grad_rows_close, oracle_backward, oracle_forward  # noqa: E402
from tests.test_gpu_edge_cases import _clustered  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
lib = _native.load()
ri = lambda a, b: int(torch.randint(a, b, (1,), generator=g).item())  # noqa: E731
seen = {4: 0, 8: 0, 16: 0}
noise = []
for it in range(n):
    W, H = 16 * ri(2, 24), 16 * ri(2, 18)
    C = [1, 2, 3, 4][ri(0, 4)]
    aux = bool(ri(0, 2))
    clusters = []
    for _ in range(ri(1, 5)):
        size = [ri(200, 600), ri(900, 1200), ri(1900, 2200), ri(2500, 6000)][ri(0, 4)]
        clusters.append((size, (ri(2, W - 2) + 0.3, ri(2, H - 2) + 0.6)))
    P = sum(s for s, _ in clusters) + ri(100, 20000)
    sc = _clustered(P, W, H, C, seed=7000 + it, clusters=clusters)
    _native.set_front_end([-1, 0, 1][it % 3])
    lib.splatraster_debug_set_split_max_waves(0)
    a = HipRun(sc, use_depth=aux, use_alpha=aux)
    lib.splatraster_debug_set_split_max_waves(-1)
    b = HipRun(sc, use_depth=aux, use_alpha=aux)
    lens = (a.state["ranges"][:, 1] - a.state["ranges"][:, 0]).long()
    order = torch.argsort(lens, descending=True)[:512]
    for L in lens[order].tolist():
        seen[16 if L >= 2048 else (8 if L >= 1024 else 4)] += 1
    assert torch.equal(a.color, b.color) and torch.equal(a.state["n_contrib"], b.state["n_contrib"]), it
    for name in ("means3D", "means2D", "opacities", "colors", "scales", "rotations"):
        ga, gb = getattr(a, name).grad.cpu().numpy(), getattr(b, name).grad.cpu().numpy()
        assert np.isfinite(gb).all(), (it, name)
        try:
            assert_grad_close(f"{it} {name}", gb, ga, rtol=1e-4, atol_scale=5e-5)
            one = ga.reshape(ga.shape[0], -1).shape[1] == 1
            assert_grad_rows_close(f"{it} rows {name}", gb, ga, rtol=1e-4, row_atol=1e-3, allow_frac=2e-3, outlier_factor=float("inf") if one else 300.0)
        except AssertionError as ex:
            # the two float32 walks disagree beyond the soak's (strict) bars: the ORACLE (double accumulation) decides — the split backward
            # must meet the parity tests' bar against it, and must not be further from it than the one-wave walk by more than that bar
            key = {"means3D": "dL_dmeans3D", "means2D": None, "opacities": "dL_dopacities", "colors": "dL_dcolors", "scales": "dL_dscales",
                   "rotations": "dL_drotations"}[name]
            if key is None:
                raise
            f = oracle_forward(sc)
            bo = oracle_backward(f, sc, use_depth=aux, use_alpha=aux)
            go = np.asarray(bo[key], dtype=np.float64).reshape(gb.shape)
            # (a few elements may miss it by a bounded factor in EITHER walk: a Gaussian of radius 90 over dozens of tiles collects hundreds of
            #  float atomics of cancelling terms, in an order that changes from run to run — scene 182 of seed 1: the same element is the worst
            #  of both walks, 2e-7 .. 1.4e-6 from the oracle depending on the run)
            assert_grad_close(f"{it} {name} split vs ORACLE", gb, go, allow_frac=1e-3, outlier_factor=30.0)
            ea, eb = np.abs(ga - go).max(), np.abs(gb - go).max()
            noise.append((it, name, str(ex)[:90], f"max err vs oracle: one-wave {ea:.2e}, split {eb:.2e}, scale {np.abs(go).max():.2e}"))
_native.set_front_end(-1)
for ev in noise:
    print("beyond the strict bars, decided by the oracle:", ev)
print(f"soak_split ok: {n} scenes; lists by part count among each frame's 512 longest: {seen}; {len(noise)} tensor(s) decided by the oracle")
