"""Per-phase cycle counters of the wide backward (SR_BWD_PROFILE, kept in tools/patches/r05_variants.patch):
    python tools/ablate.py --patch tools/patches/r05_variants.patch bprof "-DSR_BWD_PROFILE"   (here, before gpurun)
    python tools/bwd_prof.py [variant name = bprof]                                              (on the GPU box)"""
import ctypes as C, os, sys, torch
sys.path.insert(0, ".")
os.environ["SPLATRASTER_LIB"] = os.path.abspath("splatloc_amd/_lib/variants/libsplatraster_%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "bprof"))
from splatloc_amd import _native
from splatloc_amd.synthetic import make_workload
from tests.helpers import HipRun
lib = _native.load()
sc = make_workload("S2")
HipRun(sc, backward=True)
raw = C.CDLL(os.environ["SPLATRASTER_LIB"])
out = (C.c_ulonglong * 12)()
raw.splatraster_debug_bwd_prof(out, 1)
HipRun(sc, backward=True)
torch.cuda.synchronize()
raw.splatraster_debug_bwd_prof(out, 1)
v = list(out)
waves = len(range(0, 8160 * 4, 61))   # the probe samples every 61st workgroup
names = ["fetch-wait", "stage", "alpha-eval", "chain+moments", "butterfly|reduce_e", "park", "flush", "mfma-dot"]
print("per-wave avg cycles:", {n: round(v[i] / waves) for i, n in enumerate(names)}, "| whole kernel", round(v[11] / waves), "| sum of phases", round(sum(v[:8]) / waves))
print("per-wave counts: chunks %.1f rounds %.1f hit pairs %.1f" % (v[8] / waves, v[9] / waves, v[10] / waves))
print("cycles per hit pair: alpha %.0f chain %.0f butterfly %.0f atomic+park %.0f" % tuple(v[i] / max(v[10], 1) for i in (2, 3, 4, 5)))
