"""Builds libsplatraster.so (the C ABI of include/splatraster.h) with hipcc for gfx950.

Pure hipcc: no torch headers, no pybind — the library is loaded with ctypes
(splatloc_amd/_native.py).  Objects are cached by content hash (sources, headers, flags; _lib/manifest.json); the .so is kept in-tree
(splatloc_amd/_lib/) so it travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import hashlib
import json
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "_lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_PATH = os.path.join(LIB_DIR, "libsplatraster.so")
MANIFEST = os.path.join(LIB_DIR, "manifest.json")
# experiments only: SPLATRASTER_LIB points the loader at a variant build (tools/ablate.py)


ARCH = "gfx950"
# -fno-slp-vectorize: the SLP vectoriser turns pairs of fp32 operations into v_pk_{mul,add,fma}_f32,
# which measured SLOWER than the scalar pairs in both compositing kernels on MI355X (A/B in one
# run: forward 0.423 -> 0.411 ms, backward 0.989 -> 0.966 ms with it off)
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-munsafe-fp-atomics", "-fno-slp-vectorize",
          "-Wall", "-Wno-unused-function"]
# translation units whose fp32 results must round exactly like the CPU oracle
# (integer outputs derived from them are compared bit-for-bit)
# extra defines for experiments (tools/ablate.py sets this before build(force=True) into another LIB_DIR)
EXTRA_FLAGS: list = []
NO_CONTRACT = {"preprocess.hip", "binning.hip", "knn.hip"}
SOURCES = ["preprocess.hip", "preprocess_bwd.hip", "scan_sort.hip", "binning.hip", "composite_fwd.hip",
           "composite_bwd.hip", "knn.hip", "activations.hip", "losses.hip", "densify.hip", "capi.hip"]


def have_hipcc() -> bool:
    return bool(shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"))


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def _sha(paths) -> str:
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _headers():
    hdrs = sorted(os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h"))
    return hdrs + [os.path.join(_HERE, "..", "include", "splatraster.h")]


def _flags(src: str):
    flags = list(COMMON) + list(EXTRA_FLAGS)
    if src in NO_CONTRACT:
        flags.append("-ffp-contract=off")
    return flags


def _digest(src: str) -> str:
    """content hash of everything one object depends on (source, headers, flags) — not mtimes:
    the in-tree objects travel with repo snapshots whose timestamps mean nothing"""
    return _sha([os.path.join(CSRC, src)] + _headers()) + "|" + " ".join(_flags(src))


def _load_manifest() -> dict:
    try:
        with open(MANIFEST) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def _compile_one(src: str, force: bool, manifest: dict) -> tuple:
    obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
    path = os.path.join(CSRC, src)
    dig = _digest(src)
    if not force and os.path.exists(obj) and manifest.get(src) == dig:
        return obj, dig, False
    cmd = [_hipcc(), *_flags(src), "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj, dig, True


def up_to_date() -> bool:
    """True when the in-tree .so was built from exactly the present sources and flags."""
    m = _load_manifest()
    return os.path.exists(LIB_PATH) and all(m.get(s) == _digest(s) for s in SOURCES)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    if not force and up_to_date():
        if verbose:
            print("up to date", LIB_PATH)
        return LIB_PATH
    manifest = _load_manifest()
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        res = list(ex.map(lambda s: _compile_one(s, force, manifest), SOURCES))
    objs = [r[0] for r in res]
    cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB_PATH]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(MANIFEST, "w") as f:
        json.dump({s: r[1] for s, r in zip(SOURCES, res)}, f, indent=1)
    if verbose:
        print("built", LIB_PATH, "(recompiled:", [s for s, r in zip(SOURCES, res) if r[2]], ")")
    return LIB_PATH


if __name__ == "__main__":
    import sys

    build(force="--force" in sys.argv, verbose=True)
