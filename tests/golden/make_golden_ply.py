"""Generates tests/golden/ply.npz with the reference's own GaussianModel.save_ply / load_ply
(gaussian_model.py:327-377, 394-475) in THIS container.  `plyfile` (0.8.1 in environment.yml) is not
installed, so the two calls the reference makes into it are served by a RECORDER: `PlyElement.describe`
captures the structured array save_ply assembled (property names, order, values), `PlyData.read` serves
that same table back to load_ply, whose resulting tensors are recorded.  What is pinned is therefore the
reference's column layout, transposes and shapes; the byte encoding on disk is plyfile's published
binary_little_endian format, restated in splatloc_amd/ply.py.  Two models: SH degree 0 (SplatLoc's
configuration, f_rest [P,0,3]) and SH degree 1.  Only the fixture (data) is committed.
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

REC = {}


class PlyElement:
    def __init__(self, data, name):
        self.data, self.name = data, name
        self.properties = [types.SimpleNamespace(name=n) for n in data.dtype.names]

    @staticmethod
    def describe(data, name):
        REC["elements"] = data.copy()
        return PlyElement(data, name)

    def __getitem__(self, k):
        return self.data[k]


class PlyData:
    def __init__(self, elements):
        self.elements = list(elements)

    def write(self, path):
        REC["path"] = path

    @staticmethod
    def read(path):
        return PlyData([PlyElement(REC["elements"], "vertex")])

    def __getitem__(self, k):
        assert k == "vertex"
        return self.elements[0]


def main():
    for m in ("cv2", "open3d", "tinycudann", "models"):
        mg.stub(m)
    mg.stub("plyfile", PlyData=PlyData, PlyElement=PlyElement)
    mg.stub("models.decoders", FeatureDecoder=object)
    out = {}
    with mg.CudaToCpu():
        import gaussian_splatting.scene.gaussian_model as gmod
        gmod.mkdir_p = lambda p: None
        for deg in (0, 1):
            g = torch.Generator().manual_seed(40 + deg)
            P, K = 257, (deg + 1) ** 2
            gm = gmod.GaussianModel(deg, config={"Training": {"primitive_reg": True}})
            par = lambda t: torch.nn.Parameter(t.requires_grad_(True))  # noqa: E731
            gm._xyz = par(torch.randn(P, 3, generator=g))
            gm._features_dc = par(torch.randn(P, 1, 3, generator=g))
            gm._features_rest = par(torch.randn(P, K - 1, 3, generator=g))
            gm._opacity = par(torch.randn(P, 1, generator=g))
            gm._scaling = par(torch.randn(P, 3, generator=g))
            gm._rotation = par(torch.randn(P, 4, generator=g))
            gm._marker = par(torch.rand(P, 1, generator=g))
            gm._kp_score = par(torch.rand(P, 1, generator=g))
            pre = f"deg{deg}_"
            for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "_marker", "_kp_score"):
                out[pre + "in" + k] = getattr(gm, k).detach().numpy().copy()
            gm.save_ply("/nonexistent/point_cloud/final/point_cloud.ply")
            el = REC["elements"]
            out[pre + "names"] = np.array(el.dtype.names)
            out[pre + "dtypes"] = np.array([el.dtype[n].str for n in el.dtype.names])
            out[pre + "table"] = np.stack([el[n] for n in el.dtype.names], axis=1)
            gm2 = gmod.GaussianModel(deg, config={"Training": {"primitive_reg": True}})
            gm2.load_ply("ignored")
            for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "_marker", "_kp_score"):
                out[pre + "out" + k] = getattr(gm2, k).detach().numpy().copy()
            out[pre + "active_sh_degree"] = np.array(gm2.active_sh_degree)
            out[pre + "max_radii2D"] = gm2.max_radii2D.numpy().copy()
    path = os.path.join(HERE, "ply.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), list(out["deg0_names"]))


if __name__ == "__main__":
    main()
