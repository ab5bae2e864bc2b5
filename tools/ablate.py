#!/usr/bin/env python3
"""Builds variant libraries with extra -D flags for within-run A/B perf experiments.
usage: ablate.py NAME "-DFLAG=1 ..."   ->  splatloc_amd/_lib/variants/libsplatraster_NAME.so
Run a variant with SPLATRASTER_LIB=<that path> python bench.py ...
"""
import os
import subprocess
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from splatloc_amd import build as B  # noqa: E402

name, flags = sys.argv[1], sys.argv[2].split()
out_dir = os.path.join(B.LIB_DIR, "variants")
obj_dir = os.path.join(out_dir, "obj_" + name)
os.makedirs(obj_dir, exist_ok=True)
objs = []
procs = []
for src in B.SOURCES:
    obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
    fl = list(B.COMMON) + flags + (["-ffp-contract=off"] if src in B.NO_CONTRACT else [])
    procs.append((src, subprocess.Popen([B._hipcc(), *fl, "-c", os.path.join(B.CSRC, src), "-o", obj],
                                        stderr=subprocess.PIPE, text=True)))
    objs.append(obj)
for src, p in procs:
    _, err = p.communicate()
    if p.returncode:
        raise SystemExit(f"{src}: {err}")
lib = os.path.join(out_dir, f"libsplatraster_{name}.so")
subprocess.run([B._hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", *objs, "-o", lib], check=True)
print(lib)
