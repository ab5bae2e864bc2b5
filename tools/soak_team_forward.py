#!/usr/bin/env python3
"""Soak of the narrow-layout forward's team mode (composite_fwd.hip): random scenes (1 - 4 channels, 8 - 400 x 8 - 300 pixels, 1 - 30 000
Gaussians, five scale classes, translucent or not) rendered with one wave per quadrant and with a team for each of the 128 longest
lists; images, depth, alpha, n_contrib and final_T must be bit-identical.  The roles of a team meet at barriers and exchange values
through LDS: a race would show up here as a rare mismatch.     usage: python tools/soak_team_forward.py [seed=0] [scenes=150]"""
import sys, torch, numpy as np
sys.path.insert(0, ".")
from splatloc_amd import _native
from splatloc_amd.synthetic import make_scene
from tests.helpers import HipRun
lib = _native.load()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 150
bad = 0
for it in range(N):
    C = int(rng.integers(1, 5)); W = int(rng.integers(8, 400)); H = int(rng.integers(8, 300))
    P = int(rng.integers(1, 30000)); sm = float(rng.choice([0.01, 0.03, 0.08, 0.2, 0.5]))
    sc = make_scene(P, W, H, C, seed=int(rng.integers(1 << 30)), scale_median=sm)
    if rng.random() < 0.5:
        sc.opacities = sc.opacities * float(rng.choice([0.02, 0.1, 0.5]))
    outs = {}
    for mode in (0, 2):
        lib.splatraster_debug_set_fwd_team(mode)
        r = HipRun(sc, backward=False)
        outs[mode] = (r.color.clone(), r.depth.clone(), r.alpha.clone(), r.state["n_contrib"].clone(), r.state["final_T"].clone())
    ok = all(torch.equal(a, b) for a, b in zip(outs[0], outs[2]))
    if not ok:
        bad += 1
        print("MISMATCH", it, C, W, H, P, sm, flush=True)
lib.splatraster_debug_set_fwd_team(-1)
print("soak done", N, "mismatches", bad)
