"""Wave start/end timeline of the compositing kernels (100 MHz s_memrealtime): resident waves
over time, life-time distribution, tail.  Needs the debug variant:
    python tools/ablate.py --patch tools/patches/r05_variants.patch trace "-DSR_TRACE_WAVES=1"   (here, before gpurun;
                                                        the trace hooks live in the patch, not in the shipped kernels)
    python tools/wave_trace.py                          (on the GPU box)"""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
os.environ["SPLATRASTER_LIB"] = os.path.abspath("splatloc_amd/_lib/variants/libsplatraster_trace.so")
from splatloc_amd import _native
from splatloc_amd.synthetic import make_workload
from tests.helpers import HipRun
_native.load()
sc = make_workload(sys.argv[1] if len(sys.argv) > 1 else "S2")
V = int(sys.argv[2]) if len(sys.argv) > 2 else 1      # > 1: a window of V views as one launch sequence
if V > 1:
    from tests.test_gpu_window import _views, _window
    views = _views(sc, V, "cuda:0")
    _window(sc, views, "cuda:0")
    _window(sc, views, "cuda:0")
else:
    HipRun(sc, backward=True)
    HipRun(sc, backward=True)
torch.cuda.synchronize()
raw = C.CDLL(os.environ["SPLATRASTER_LIB"])
n = V * ((sc.camera.image_width + 15) // 16) * ((sc.camera.image_height + 15) // 16)
n = (n + 7) // 8 * 32
assert n <= 40960, "trace buffer"
for k in ("fwd", "bwd"):
    buf = (C.c_ulonglong * (2 * n))()
    assert getattr(raw, "splatraster_debug_trace_" + k)(buf, n) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 2).astype(np.int64)
    t0, t1 = a[:, 0], a[:, 1]
    ok = t1 > 0
    t0, t1 = t0[ok], t1[ok]
    base = t0.min()
    s, e = (t0 - base) / 100.0, (t1 - base) / 100.0      # us
    life = e - s
    T = e.max()
    print(f"== {k}: {ok.sum()} waves, kernel span {T:.0f} us; life us: mean {life.mean():.1f} p10 {np.percentile(life,10):.1f} "
          f"p50 {np.percentile(life,50):.1f} p90 {np.percentile(life,90):.1f} max {life.max():.1f}")
    # resident waves over time in 20 buckets
    edges = np.linspace(0, T, 21)
    res = [((s < edges[i + 1]) & (e > edges[i])).sum() for i in range(20)]
    started = [((s >= edges[i]) & (s < edges[i + 1])).sum() for i in range(20)]
    print("   resident per 5% of the span:", [int(r) for r in res])
    print("   waves started per 5% of the span:", [int(r) for r in started])
    print(f"   last wave starts at {s.max():.0f} us; time with < 50% of peak residency: "
          f"{sum(1 for r in res if r < 0.5 * max(res)) * 5}% of the span")
    # what would a longest-first launch order buy?  list scheduling of the measured life times on the slots seen at the start
    import heapq
    slots = int(res[0])
    def makespan(order):
        h = [0.0] * slots
        heapq.heapify(h)
        end = 0.0
        for L in order:
            t = heapq.heappop(h)
            heapq.heappush(h, t + L)
            end = max(end, t + L)
        return end
    blk = np.nonzero(ok)[0]
    life_by_block = life[np.argsort(blk, kind="stable")]
    print(f"   list-scheduling model on {slots} slots: launch order {makespan(life_by_block):.0f} us, longest first "
          f"{makespan(np.sort(life_by_block)[::-1]):.0f} us, ideal {life.sum() / slots:.0f} us")
