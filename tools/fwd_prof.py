"""Per-phase cycle counters of the wide forward (timing probe: tools/patches/r04_fwd_profile.patch, built as a variant library with
`python tools/ablate.py --patch tools/patches/r04_fwd_profile.patch fprof "-DSR_FWD_PROFILE"`).  usage: python tools/fwd_prof.py [variant]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, ".")
os.environ["SPLATRASTER_LIB"] = os.path.abspath("splatloc_amd/_lib/variants/libsplatraster_%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "fprof"))
from splatloc_amd import _native
from splatloc_amd.synthetic import make_workload
from tests.helpers import HipRun
lib = _native.load()
sc = make_workload("S2")
HipRun(sc, backward=False)
raw = C.CDLL(os.environ["SPLATRASTER_LIB"])
out = (C.c_ulonglong * 12)()
raw.splatraster_debug_fwd_prof(out, 1)
HipRun(sc, backward=False)
torch.cuda.synchronize()
raw.splatraster_debug_fwd_prof(out, 1)
v = list(out)
waves = len(range(0, 8160 * 4, 61))   # the probe samples every 61st workgroup
names = ["start-up", "chunk hand-over", "staging (+ waits)", "pair loops", "epilogue"]
print("per-wave avg cycles (100 MHz s_memtime ticks x clock ratio: see readcyclecounter):", {n: round(v[i] / waves) for i, n in enumerate(names)},
      "| whole kernel", round(v[11] / waves), "| sum of phases", round(sum(v[:5]) / waves))
print("per-wave counts: chunks %.1f rounds %.1f pairs %.1f" % (v[8] / waves, v[9] / waves, v[10] / waves))
print("cycles per pair %.0f, per staging round %.0f; epilogue: %.0f until the last store is issued, %.0f waiting for the stores" % (
    v[3] / max(v[10], 1), v[2] / max(v[9], 1), v[5] / waves, (v[4] - v[5]) / waves))
print("lane utilisation: %.1f of 64 pixels hit per evaluated candidate (%.1f %%); %.1f candidates per wave" % (
    v[6] / max(v[7], 1), 100.0 * v[6] / max(64 * v[7], 1), v[7] / waves))
