import sys, os, types, torch, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import map_idle
from splatloc_amd import _native, training
dev = torch.device("cuda:0")
pc, views = map_idle.build("S0", dev)
bg = torch.zeros(3, device=dev)
pipe = types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False)
cfg = {"Training": {"rgb_boundary_threshold": 0.01, "primitive_reg": True}}
_native.set_deterministic(True)
for with_reg in (False, True):
    res = {}
    for raw in (False, True):
        training.RAW_BACKWARD = raw
        for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_kp_score", "_scaling", "_rotation"):
            getattr(pc, k).grad = None
        for cam in views: cam.exposure_a.grad = cam.exposure_b.grad = None
        training._map_grads_direct(views, pc, pipe, bg, cfg, with_reg)
        torch.cuda.synchronize()
        res[raw] = {k: getattr(pc, k).grad.detach().clone() for k in ("_scaling", "_rotation", "_opacity")}
    for k in res[False]:
        a, b = res[False][k], res[True][k]
        d = a.view(torch.int32) != b.view(torch.int32)
        print("with_reg", with_reg, k, "differing", int(d.sum()), "of", d.numel(), "max abs", float((a - b).abs().max()))
    if with_reg:
        a, b = res[False]["_scaling"], res[True]["_scaling"]
        idx = (a.view(torch.int32) != b.view(torch.int32)).nonzero()[:5]
        print(idx.tolist(), [(float(a[i, j]), float(b[i, j]), float(pc._marker[i])) for i, j in idx.tolist()])
