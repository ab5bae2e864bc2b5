import sys, os, torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import map_idle
from splatloc_amd import training
from splatloc_amd.fused import _ActivatePack, _view_settings
from splatloc_amd.rasterizer import PlainCtx, _RasterizeWindow
dev = torch.device("cuda:0")
pc, views = map_idle.build("S0", dev)
bg = torch.zeros(3, device=dev)
with torch.no_grad():
    xyz = pc._xyz
    c_act = PlainCtx()
    A = _ActivatePack.forward(c_act, xyz, pc._features_dc, pc._features_rest, pc._scaling, pc._rotation, pc._opacity, pc._kp_score, None, 0)
    P = xyz.shape[0]; f32 = dict(dtype=torch.float32, device=dev)
    B = (torch.empty((P, 3), **f32), torch.empty((P, 4), **f32), torch.empty((P, 1), **f32), torch.empty((P, 4), **f32))
    c = PlainCtx()
    c.raw_fwd = (pc._scaling.detach(), pc._rotation.detach(), pc._opacity.detach(), pc._features_dc.detach(), pc._kp_score.detach())
    st = _view_settings(views[1], pc, bg, 1.0)
    outB = _RasterizeWindow.forward(c, xyz, B[3], B[2], B[0], B[1], None, (st,), 3, None, xyz)
    c2 = PlainCtx()
    outA = _RasterizeWindow.forward(c2, xyz, A[3], A[2], A[0], A[1], None, (st,), 3, None, xyz)
    torch.cuda.synchronize()
    for name, a, b in zip(("scales", "rotations", "opacities", "colors"), A, B):
        d = (a.view(torch.int32) != b.view(torch.int32))
        print(name, "differing elements", int(d.sum()), "of", d.numel(), "max abs diff", float((a - b).abs().max()))
    print("images equal", torch.equal(outA[0], outB[0]), "radii equal", torch.equal(outA[4], outB[4]))
