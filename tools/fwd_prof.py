import ctypes as C, os, sys, torch
sys.path.insert(0, ".")
os.environ["SPLATRASTER_LIB"] = os.path.abspath("splatloc_amd/_lib/variants/libsplatraster_fprof.so")
from splatloc_amd import _native
from splatloc_amd.synthetic import make_workload
from tests.helpers import HipRun
lib = _native.load()
sc = make_workload("S2")
HipRun(sc, backward=False)
raw = C.CDLL(os.environ["SPLATRASTER_LIB"])
out = (C.c_ulonglong * 8)()
raw.splatraster_debug_fwd_prof(out, 1)
HipRun(sc, backward=False)
torch.cuda.synchronize()
raw.splatraster_debug_fwd_prof(out, 1)
v = list(out)
waves = 8160 * 4
names = ["fetch", "stage", "composite", "epilogue"]
tot = sum(v[:4])
print("per-wave avg cycles:", {n: round(v[i] / waves) for i, n in enumerate(names)}, "total", round(tot / waves))
print("per-wave counts: chunks %.1f rounds %.1f cands %.1f hits %.1f" % tuple(x / waves for x in v[4:]))
print("cycles per: chunk fetch %.0f, stage round %.0f, candidate %.0f" % (v[0] / max(v[4], 1), v[1] / max(v[5], 1), v[2] / max(v[6], 1)))
