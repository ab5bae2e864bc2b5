import sys, time, cProfile, pstats, io
sys.path.insert(0, ".")
import torch
from splatloc_amd.synthetic import make_scene
from splatloc_amd import GaussianRasterizer
from tests.helpers import hip_settings
dev = torch.device("cuda:0")
sc = make_scene(2000, 160, 96, 4, 3, scale_median=0.03).to(dev)
leaf = lambda t: t.clone().requires_grad_(True)
m3, col, op, sca, rot = leaf(sc.means3D), leaf(sc.features), leaf(sc.opacities), leaf(sc.scales), leaf(sc.rotations)
rast = GaussianRasterizer(raster_settings=hip_settings(sc, dev))
params = [m3, col, op, sca, rot]
def fwd():
    m2 = torch.zeros_like(m3, requires_grad=True)
    return rast(means3D=m3, means2D=m2, shs=None, colors_precomp=col, opacities=op, scales=sca, rotations=rot, cov3D_precomp=None)
def step():
    for p in params: p.grad = None
    c, d, a, r = fwd()
    torch.autograd.backward((c, d, a), (sc.dL_dcolor, sc.dL_ddepth, sc.dL_dalpha))
for _ in range(50): step()
torch.cuda.synchronize()
N = 1000
t0 = time.perf_counter()
for _ in range(N):
    with torch.no_grad(): fwd()
torch.cuda.synchronize(); t1 = time.perf_counter()
for _ in range(N): step()
torch.cuda.synchronize(); t2 = time.perf_counter()
print("fwd-only (no_grad) ms", (t1 - t0) / N * 1e3, " fwd+bwd ms", (t2 - t1) / N * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(500): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats(22); print(s.getvalue()[:4500])
