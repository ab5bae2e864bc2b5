"""Generates tests/golden/reset_opacity.npz with the reference's own GaussianModel.reset_opacity_nonvisible
(gaussian_model.py:384-392 -> replace_tensor_to_optimizer :477-490) in THIS container: opacity logits and the
opacity group's Adam state before / after, for two visibility filters.  Only the fixture is committed."""
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
    out = {}
    with mg.CudaToCpu():
        from gaussian_splatting.scene.gaussian_model import GaussianModel
        g = torch.Generator().manual_seed(321)
        P = 500
        gm = GaussianModel(0, config={"Training": {"primitive_reg": False}})
        par = lambda t: torch.nn.Parameter(t.contiguous().requires_grad_(True))  # noqa: E731
        gm._xyz, gm._features_dc, gm._features_rest = par(torch.randn(P, 3, generator=g)), par(torch.rand(P, 1, 3, generator=g)), par(torch.zeros(P, 0, 3))
        gm._opacity, gm._marker, gm._kp_score = par(torch.randn(P, 1, generator=g) * 2), par(torch.rand(P, 1, generator=g)), par(torch.rand(P, 1, generator=g))
        gm._scaling, gm._rotation = par(torch.randn(P, 3, generator=g)), par(torch.randn(P, 4, generator=g))
        args = types.SimpleNamespace(percent_dense=0.01, position_lr_init=0.0016, position_lr_final=0.0000016,
                                     position_lr_delay_mult=0.01, position_lr_max_steps=30000, feature_lr=0.0025,
                                     opacity_lr=0.05, marker_lr=0.05, kp_score_lr=0.05, scaling_lr=0.001, rotation_lr=0.001)
        gm.spatial_lr_scale = 6.0
        gm.training_setup(args)
        for _ in range(2):
            gm._opacity.grad = torch.randn(P, 1, generator=g) * 1e-3
            gm.optimizer.step()
        out["opacity_before"] = gm._opacity.detach().numpy().copy()
        st = gm.optimizer.state[gm._opacity]
        out["m_before"], out["v_before"], out["step_before"] = st["exp_avg"].numpy().copy(), st["exp_avg_sq"].numpy().copy(), np.array(float(st["step"]))
        f0 = torch.rand(P, generator=g) < 0.4
        f1 = torch.rand(P, generator=g) < 0.3
        out["filter0"], out["filter1"] = f0.numpy().copy(), f1.numpy().copy()
        gm.reset_opacity_nonvisible([f0, f1])
        out["opacity_after"] = gm._opacity.detach().numpy().copy()
        st = gm.optimizer.state[gm._opacity]
        out["m_after"], out["v_after"], out["step_after"] = st["exp_avg"].numpy().copy(), st["exp_avg_sq"].numpy().copy(), np.array(float(st["step"]))
        assert gm.optimizer.param_groups[3]["params"][0] is gm._opacity
    path = os.path.join(HERE, "reset_opacity.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
