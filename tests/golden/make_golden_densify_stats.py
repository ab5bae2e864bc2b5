"""Generates tests/golden/densify_stats.npz with the reference's own code in THIS container:
GaussianModel.add_densification_stats (gaussian_model.py:677-679) and the max_radii2D update of
the mapping loop (train_gaussians.py:238-245), applied for 3 consecutive views to seeded state.
Only the fixture is committed.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    for m in ("cv2", "open3d", "tinycudann", "models"):
        mg.stub(m)
    mg.stub("plyfile", PlyData=object, PlyElement=object)
    mg.stub("models.decoders", FeatureDecoder=object)
    with mg.CudaToCpu():
        from gaussian_splatting.scene.gaussian_model import GaussianModel
        g = torch.Generator().manual_seed(5)
        P = 777
        gm = GaussianModel(0, config={"Training": {"primitive_reg": True}})
        gm.xyz_gradient_accum = torch.rand(P, 1, generator=g)
        gm.denom = torch.randint(0, 4, (P, 1), generator=g).float()
        gm.max_radii2D = torch.randint(0, 30, (P,), generator=g).float()
        out = {"accum0": gm.xyz_gradient_accum.numpy().copy(), "denom0": gm.denom.numpy().copy(),
               "max_radii0": gm.max_radii2D.numpy().copy()}
        for v in range(3):
            grad = torch.randn(P, 3, generator=g) * 1e-3
            radii = torch.randint(0, 40, (P,), generator=g, dtype=torch.int32)
            radii[torch.rand(P, generator=g) < 0.4] = 0
            vs = types.SimpleNamespace(grad=grad)
            vis = radii > 0
            # train_gaussians.py:240-245
            gm.max_radii2D[vis] = torch.max(gm.max_radii2D[vis], radii[vis])
            gm.add_densification_stats(vs, vis)
            out[f"grad{v}"] = grad.numpy().copy()
            out[f"radii{v}"] = radii.numpy().copy()
            out[f"accum{v + 1}"] = gm.xyz_gradient_accum.numpy().copy()
            out[f"denom{v + 1}"] = gm.denom.numpy().copy()
            out[f"max_radii{v + 1}"] = gm.max_radii2D.numpy().copy()
    path = os.path.join(HERE, "densify_stats.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
