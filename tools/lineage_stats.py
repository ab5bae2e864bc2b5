#!/usr/bin/env python3
"""What the lineage-literal comparison (tests/test_gpu_lineage_spec.py: oracle mode 1, the spec's expf form) actually measures at the
three full sizes: per gradient tensor, the fraction of elements beyond rtol 2e-3 + 1e-4 max|ref| and the worst excess factor — the
numbers the test's allowances (allow_frac, outlier_factor) are set from.       usage (GPU box): python tools/lineage_stats.py"""
import sys, numpy as np
sys.path.insert(0, ".")
from oracle import oracle
from splatloc_amd.synthetic import make_workload
from tests.helpers import HipRun, oracle_backward, oracle_forward
for name in ["S0", "S2-ref-layout", "S2"]:
    sc = make_workload(name)
    try:
        oracle.set_alpha_mode(1)
        f = oracle_forward(sc); b = oracle_backward(f, sc)
    finally:
        oracle.set_alpha_mode(0)
    run = HipRun(sc)
    for k, got in (("dL_dmeans3D", run.means3D.grad), ("dL_dmeans2D", run.means2D.grad), ("dL_dopacities", run.opacities.grad),
                   ("dL_dcolors", run.colors.grad), ("dL_dscales", run.scales.grad), ("dL_drotations", run.rotations.grad)):
        g = run.np(got).astype(np.float64); r = b[k].astype(np.float64)
        tol = 2e-3 * np.abs(r) + 1e-4 * np.abs(r).max() + 1e-30
        e = np.abs(g - r)
        bad = e > tol
        print(name, k, "bad frac %.2e" % bad.mean(), "worst factor %.1f" % (e / tol).max(), flush=True)
