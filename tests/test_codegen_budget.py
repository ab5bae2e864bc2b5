"""Register / scratch / occupancy budget of the two compositing kernels, read from the compiler's own report
(`-Rpass-analysis=kernel-resource-usage`; hipcc cross-compiles gfx950 without a GPU).  Round 4 traced two regressions of
the headline kernel to code generation nobody had looked at — a "prefetch" whose registers were spilled right behind the
loads, an epilogue with a wait between every two stores — so the numbers the design depends on (HISTORY.md §9) are
pinned here: a source or toolchain change that costs a wave per SIMD or re-introduces scratch traffic fails on the CPU."""
import os
import re
import shutil
import subprocess

import pytest

from splatloc_amd import build as B


def _usage(src):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    flags = [f for f in B._flags(src) if f not in ("-fPIC",)]
    r = subprocess.run([hipcc, *flags, "--cuda-device-only", "-c", "-Rpass-analysis=kernel-resource-usage",
                        os.path.join(B.CSRC, src), "-o", os.devnull], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out, cur = {}, None
    for ln in r.stderr.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", ln)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|SGPRs Spill|"
                      r"LDS Size \[bytes/block\]): (\d+)", ln)
        if m and cur is not None:
            cur[m.group(1).split(" [")[0]] = int(m.group(2))
    return out


def _kernel(usage, prefix):
    hits = {k: v for k, v in usage.items() if k.startswith(prefix)}
    assert len(hits) == 1, (prefix, list(usage))
    return next(iter(hits.values()))


def test_forward_kernels_have_no_scratch_and_keep_their_occupancy():
    u = _usage("composite_fwd.hip")
    fwd = {int(re.search(r"composite_fwd_kernelILi(\d+)E", k).group(1)): v for k, v in u.items() if "composite_fwd_kernel" in k}
    narrow = {int(re.search(r"composite_fwd_narrow_kernelILi(\d+)E", k).group(1)): v for k, v in u.items() if "composite_fwd_narrow_kernel" in k}
    mixed = {int(re.search(r"composite_fwd_mixed_kernelILi(\d+)E", k).group(1)): v for k, v in u.items() if "composite_fwd_mixed_kernel" in k}
    assert sorted(fwd) == [8, 16, 32, 35] and sorted(narrow) == [1, 2, 3, 4] and sorted(mixed) == [1, 2, 3, 4]
    for nc, k in list(fwd.items()) + list(narrow.items()) + list(mixed.items()):
        assert k["ScratchSize"] == 0 and k["VGPRs Spill"] == 0, (nc, k)
    for nc in (1, 2, 3, 4):
        # the declared budget, __launch_bounds__(WAVE, 7): <= 72 registers, >= 7 waves per SIMD (today's toolchain: 64 / 8)
        assert narrow[nc]["VGPRs"] <= 72 and narrow[nc]["Occupancy"] >= 7, (nc, narrow[nc])
        # one workgroup of four waves per tile / per quadrant of a long list: five workgroups per CU by LDS, never fewer by registers
        assert mixed[nc]["VGPRs"] <= 96 and mixed[nc]["Occupancy"] >= 5 and mixed[nc]["LDS Size"] <= 28 * 1024, (nc, mixed[nc])
    assert fwd[35]["VGPRs"] <= 96 and fwd[35]["Occupancy"] >= 5, fwd[35]      # the headline layout: 5 waves per SIMD
    assert fwd[35]["LDS Size"] <= 5700, fwd[35]                               # 28 workgroups per CU by LDS


def test_backward_kernels_stay_within_their_register_budget():
    u = _usage("composite_bwd.hip")
    wide = _kernel(u, "_ZN2sr20composite_bwd_kernelILi35ELb0ELb0ELb1E")        # <35, normal mode, butterfly variant, with depth / alpha>
    assert wide["VGPRs"] <= 128 and wide["Occupancy"] == 4, wide
    assert wide["VGPRs Spill"] <= 3 and wide["ScratchSize"] <= 12, wide        # (the spilled values live outside the loops: HISTORY.md §9)
    for nc in (1, 2, 3):                                                        # the butterfly variants of the narrow layouts: full occupancy, no scratch
        k = _kernel(u, f"_ZN2sr20composite_bwd_kernelILi{nc}ELb0ELb0ELb1E")
        assert k["Occupancy"] == 8 and k["ScratchSize"] == 0, (nc, k)
    ref = _kernel(u, "_ZN2sr20composite_bwd_kernelILi4ELb0ELb1ELb1E")          # SplatLoc's own layout, small-panel variant
    assert ref["ScratchSize"] == 0 and ref["Occupancy"] >= 5, ref
    refine = _kernel(u, "_ZN2sr20composite_bwd_kernelILi3ELb0ELb0ELb0E")       # color_refinement: 3 channels, no depth / alpha terms
    assert refine["ScratchSize"] == 0 and refine["Occupancy"] == 8, refine
