#!/usr/bin/env python3
"""Builds variant libraries with extra -D flags for within-run A/B perf experiments.
usage: ablate.py [--patch tools/patches/X.patch [--patch Y.patch ...]] NAME "-DFLAG=1 ..."   ->  splatloc_amd/_lib/variants/libsplatraster_NAME.so
Run a variant with SPLATRASTER_LIB=<that path> python bench.py ...

--patch: the sources are copied to a scratch directory and the patch is applied there first (patch -p1).  The timing probes
that produce WRONG results by design (dropped atomics / butterfly / MFMAs / plane loads, hot-row gathers, skipped staging:
-DSR_BWD_PROBE=n, -DSR_ABLATE_*, -DSR_BWD_ABLATE_ATOMIC, -DSR_FWD_PROBE=1) are kept as tools/patches/composite_probes.patch
instead of living in the shipped translation units; their results are in profiles/r03_ab_probes.txt.
"""
import os
import subprocess
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from splatloc_amd import build as B  # noqa: E402

argv = sys.argv[1:]
patches = []
while argv and argv[0] == "--patch":
    patches.append(os.path.abspath(argv[1]))
    argv = argv[2:]
patch = patches[0] if patches else None
name, flags = argv[0], (argv[1].split() if len(argv) > 1 else [])
CSRC = B.CSRC
if patch:
    import shutil
    import tempfile
    root = tempfile.mkdtemp(prefix="splat_probe_")
    shutil.copytree(os.path.dirname(B.CSRC), os.path.join(root, "splatloc_amd"), ignore=shutil.ignore_patterns("_lib", "__pycache__"))
    shutil.copytree(os.path.join(os.path.dirname(os.path.dirname(B.CSRC)), "include"), os.path.join(root, "include"))
    for pt in patches:      # in the order given (every patch applies to the shipped tree; later ones tolerate the offsets of earlier ones)
        subprocess.run(["patch", "-p1", "-i", pt], cwd=root, check=True)
    CSRC = os.path.join(root, "splatloc_amd", "csrc")
out_dir = os.path.join(B.LIB_DIR, "variants")
obj_dir = os.path.join(out_dir, "obj_" + name)
os.makedirs(obj_dir, exist_ok=True)
objs = []
procs = []
for src in B.SOURCES:
    obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
    fl = list(B.COMMON) + flags + (["-ffp-contract=off"] if src in B.NO_CONTRACT else [])
    procs.append((src, subprocess.Popen([B._hipcc(), *fl, "-c", os.path.join(CSRC, src), "-o", obj],
                                        stderr=subprocess.PIPE, text=True)))
    objs.append(obj)
for src, p in procs:
    _, err = p.communicate()
    if p.returncode:
        raise SystemExit(f"{src}: {err}")
lib = os.path.join(out_dir, f"libsplatraster_{name}.so")
subprocess.run([B._hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", *objs, "-o", lib], check=True)
print(lib)
