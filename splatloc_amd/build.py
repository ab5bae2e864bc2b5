"""Builds libsplatraster.so (the C ABI of include/splatraster.h) with hipcc for gfx950.

Pure hipcc: no torch headers, no pybind — the library is loaded with ctypes
(splatloc_amd/_native.py).  Objects are cached by source mtime; the .so is kept in-tree
(splatloc_amd/_lib/) so it travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "_lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_PATH = os.path.join(LIB_DIR, "libsplatraster.so")
# experiments only: SPLATRASTER_LIB points the loader at a variant build (tools/ablate.py)


ARCH = "gfx950"
# -fno-slp-vectorize: the SLP vectoriser turns pairs of fp32 operations into v_pk_{mul,add,fma}_f32,
# which measured SLOWER than the scalar pairs in both compositing kernels on MI355X (A/B in one
# run: forward 0.423 -> 0.411 ms, backward 0.989 -> 0.966 ms with it off)
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-munsafe-fp-atomics", "-fno-slp-vectorize",
          "-Wall", "-Wno-unused-function"]
# translation units whose fp32 results must round exactly like the CPU oracle
# (integer outputs derived from them are compared bit-for-bit)
NO_CONTRACT = {"preprocess.hip", "binning.hip", "knn.hip"}
SOURCES = ["preprocess.hip", "preprocess_bwd.hip", "scan_sort.hip", "binning.hip", "composite_fwd.hip",
           "composite_bwd.hip", "knn.hip", "activations.hip", "losses.hip", "capi.hip"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def _deps_mtime() -> float:
    hdrs = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    hdrs += [os.path.join(_HERE, "..", "include", "splatraster.h"), os.path.abspath(__file__)]
    return max(os.path.getmtime(h) for h in hdrs)


def _compile_one(src: str, force: bool) -> str:
    obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
    path = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj)
            and os.path.getmtime(obj) >= max(os.path.getmtime(path), _deps_mtime())):
        return obj
    flags = list(COMMON)
    if src in NO_CONTRACT:
        flags.append("-ffp-contract=off")
    cmd = [_hipcc(), *flags, "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(lambda s: _compile_one(s, force), SOURCES))
    if (force or not os.path.exists(LIB_PATH)
            or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(o) for o in objs)):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB_PATH]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print("built", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    import sys

    build(force="--force" in sys.argv, verbose=True)
