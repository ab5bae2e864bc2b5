"""Builds libsplatraster.so (the C ABI of include/splatraster.h) with hipcc for gfx950.

Pure hipcc: no torch headers, no pybind — the library is loaded with ctypes
(splatloc_amd/_native.py).  Objects are cached by content hash (sources, headers, flags; _lib/manifest.json); the .so is kept in-tree
(splatloc_amd/_lib/) so it travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import contextlib
import fcntl
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
LOCK_PATH = os.path.join(LIB_DIR, ".build.lock")
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
NO_CONTRACT = {"preprocess.hip", "binning.hip", "binsort.hip", "knn.hip"}
SOURCES = ["preprocess.hip", "preprocess_bwd.hip", "scan_sort.hip", "binning.hip", "binsort.hip", "composite_fwd.hip",
           "composite_bwd.hip", "knn.hip", "activations.hip", "losses.hip", "densify.hip", "pose.hip", "capi.hip"]


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
    tmp = f"{obj}.{os.getpid()}.tmp"   # never a half-written object under the final name
    cmd = [_hipcc(), *_flags(src), "-c", path, "-o", tmp]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        with contextlib.suppress(OSError):
            os.remove(tmp)
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    os.replace(tmp, obj)
    return obj, dig, True


def up_to_date() -> bool:
    """True when the in-tree .so was built from exactly the present sources and flags."""
    m = _load_manifest()
    return os.path.exists(LIB_PATH) and all(m.get(s) == _digest(s) for s in SOURCES)


@contextlib.contextmanager
def _build_lock():
    """Exclusive inter-process lock on _lib/.build.lock: every rank of a `torchrun --nproc-per-node N` job imports
    this package at the same time, and after a source edit each of them would otherwise compile the same objects,
    link the same .so and write the same manifest concurrently (a rank could dlopen a half-written ELF).  The first
    rank builds; the others block here, then find everything up to date."""
    os.makedirs(LIB_DIR, exist_ok=True)
    try:
        f = open(LOCK_PATH, "a+")
    except OSError:      # read-only tree: nothing can be rebuilt there anyway
        yield
        return
    with f:
        locked = True
        try:
            fcntl.flock(f.fileno(), fcntl.LOCK_EX)
        except OSError:      # a file system without advisory locks: proceed unlocked (single-process use still works)
            locked = False
        try:
            yield
        finally:
            if locked:
                fcntl.flock(f.fileno(), fcntl.LOCK_UN)


def build(force: bool = False, verbose: bool = False) -> str:
    with _build_lock():
        return _build_locked(force, verbose)


def _build_locked(force: bool, verbose: bool) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    if not force and up_to_date():
        if verbose:
            print("up to date", LIB_PATH)
        return LIB_PATH
    manifest = _load_manifest()
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        res = list(ex.map(lambda s: _compile_one(s, force, manifest), SOURCES))
    objs = [r[0] for r in res]
    tmp_lib = f"{LIB_PATH}.{os.getpid()}.tmp"
    cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", tmp_lib]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        with contextlib.suppress(OSError):
            os.remove(tmp_lib)
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    os.replace(tmp_lib, LIB_PATH)        # atomic: a concurrent dlopen sees the old or the new library, never a torso
    tmp_manifest = f"{MANIFEST}.{os.getpid()}.tmp"
    with open(tmp_manifest, "w") as f:
        json.dump({s: r[1] for s, r in zip(SOURCES, res)}, f, indent=1)
    os.replace(tmp_manifest, MANIFEST)
    if verbose:
        print("built", LIB_PATH, "(recompiled:", [s for s, r in zip(SOURCES, res) if r[2]], ")")
    return LIB_PATH


if __name__ == "__main__":
    import sys

    build(force="--force" in sys.argv, verbose=True)
