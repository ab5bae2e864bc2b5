"""distCUDA2 timing (tiled brute force below 10 000 points, exact grid search beyond): python tools/knn_time.py > profiles/rNN_knn.json"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from simple_knn._C import distCUDA2  # noqa: E402
from splatloc_amd import _native  # noqa: E402

lib = _native.load()
rows = []
for N in (5000, 20000, 32768, 100000, 500000):
    g = torch.Generator().manual_seed(N)
    xy = torch.rand(N, 2, generator=g) * 6 - 3
    pts = torch.cat([xy, 1.5 + 0.3 * torch.sin(2 * xy[:, :1]) * torch.cos(3 * xy[:, 1:]) + 0.002 * torch.randn(N, 1, generator=g)], 1).cuda()
    row = {"N": N, "cloud": "depth-map-like surface"}
    for name, thr in (("grid_ms", 0), ("brute_ms", 1 << 30)):
        if name == "brute_ms" and N > 100000:
            continue
        lib.splatknn_debug_set_grid_min(thr)
        distCUDA2(pts)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            d = distCUDA2(pts)
        torch.cuda.synchronize()
        row[name] = round((time.perf_counter() - t0) / 5 * 1e3, 4)
    lib.splatknn_debug_set_grid_min(-1)
    row["default_path"] = "grid" if N >= 10000 else "brute force"
    row["mean_dist2"] = float(d.mean())
    rows.append(row)
print(json.dumps({"what": "simple_knn.distCUDA2 on MI355X, wall ms per call incl. workspace allocation (5 calls after 1 warm-up)", "rows": rows}, indent=1))
