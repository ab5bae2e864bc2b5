import sys, time, torch
sys.path.insert(0, ".")
from simple_knn._C import distCUDA2
for N in (5000, 20000, 100000, 300000):
    g = torch.Generator().manual_seed(N)
    pts = torch.rand(N, 3, generator=g).cuda() * 5
    distCUDA2(pts); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): d = distCUDA2(pts)
    torch.cuda.synchronize()
    print(N, "ms", (time.perf_counter() - t0) / 3 * 1e3, float(d.mean()))
