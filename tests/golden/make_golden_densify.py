"""Generates tests/golden/densify.npz with the reference's own code in THIS container:

  * torch.optim.Adam over the 8 parameter groups exactly as GaussianModel.training_setup builds them
    (gaussian_model.py:250-300: lr per group, eps 1e-15), a few steps with seeded gradients and
    `_marker.grad is None` (train_gaussians.py never gives the marker a gradient in map());
  * GaussianModel.densify_and_prune (gaussian_model.py:590-675 -> densify_and_clone, densify_and_split,
    prune_points, cat_tensors_to_optimizer / _prune_optimizer :477-587) on that model, with the
    thresholds of configs/replica_nerf/base_config.yaml;
  * two more Adam steps on the re-sized model (carried-over `step`, zero moments of the new rows).

The split's `torch.normal` draw is INJECTED, not reproduced: torch.normal is replaced for the duration
of the call by `unit[copy, source_row] * std`, where `unit` is a recorded standard-normal table indexed
by the row the child comes from — the same table the device kernel is given.  Two cases: primitive_reg
False and True (the marker gate of the final prune).  Only the fixture (data) is committed.
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

GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "marker", "kp_score", "scaling", "rotation")
ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
        "marker": "_marker", "kp_score": "_kp_score", "scaling": "_scaling", "rotation": "_rotation"}


def snapshot(gm, pre, out):
    for name in GROUPS:
        p = getattr(gm, ATTR[name])
        out[f"{pre}{name}"] = p.detach().numpy().copy()
    for grp in gm.optimizer.param_groups:
        st = gm.optimizer.state.get(grp["params"][0], None)
        assert grp["params"][0] is getattr(gm, ATTR[grp["name"]])
        if st is not None and len(st):
            out[f"{pre}m_{grp['name']}"] = st["exp_avg"].numpy().copy()
            out[f"{pre}v_{grp['name']}"] = st["exp_avg_sq"].numpy().copy()
            out[f"{pre}step_{grp['name']}"] = np.array(float(st["step"]))
        out[f"{pre}lr_{grp['name']}"] = np.array(grp["lr"])
    out[f"{pre}accum"] = gm.xyz_gradient_accum.numpy().copy()
    out[f"{pre}denom"] = gm.denom.numpy().copy()
    out[f"{pre}max_radii"] = gm.max_radii2D.numpy().copy()


def adam_steps(gm, g, n, pre, out, first_iteration):
    for k in range(n):
        grads = {}
        for name in GROUPS:
            p = getattr(gm, ATTR[name])
            if name == "marker":
                p.grad = None          # map() never back-propagates into the marker
                continue
            gr = torch.randn(p.shape, generator=g) * 1e-3
            p.grad = gr
            grads[name] = gr.numpy().copy()
        for name, a in grads.items():
            out[f"{pre}grad{k}_{name}"] = a
        gm.optimizer.step()
        gm.optimizer.zero_grad(set_to_none=True)
        lr = gm.update_learning_rate(first_iteration + k)     # train_gaussians.py:267 (after the step)
        out[f"{pre}xyz_lr_after{k}"] = np.array(lr)


def main():
    for m in ("cv2", "open3d", "tinycudann", "models"):
        mg.stub(m)
    mg.stub("plyfile", PlyData=object, PlyElement=object)
    mg.stub("models.decoders", FeatureDecoder=object)
    out = {}
    with mg.CudaToCpu():
        from gaussian_splatting.scene.gaussian_model import GaussianModel
        from gaussian_splatting.utils.general_utils import helper
        import gaussian_splatting.scene.gaussian_model as gmod
        # ---- lr schedule (general_utils.py:79-94), SURVEY §8c item 6 ----
        steps = np.array([0, 1, 100, 30000, 2_000_000, -1])
        out["lr_steps"] = steps
        out["lr_values"] = np.array([helper(int(s), lr_init=0.0016 * 6.0, lr_final=0.0000016 * 6.0, lr_delay_mult=0.01,
                                            max_steps=30000) for s in steps])
        out["lr_values_delay"] = np.array([helper(int(s), lr_init=1e-2, lr_final=1e-4, lr_delay_steps=500,
                                                  lr_delay_mult=0.01, max_steps=30000) for s in steps])
        for case, reg in (("plain_", False), ("reg_", True)):
            g = torch.Generator().manual_seed(91 if reg else 90)
            P = 640
            gm = GaussianModel(0, config={"Training": {"primitive_reg": reg}})
            extent, pd = 6.0, 0.01          # percent_dense * extent = 0.06
            par = lambda t: torch.nn.Parameter(t.contiguous().requires_grad_(True))  # noqa: E731
            gm._xyz = par(torch.randn(P, 3, generator=g) * 2.0)
            gm._features_dc = par(torch.rand(P, 1, 3, generator=g))
            gm._features_rest = par(torch.zeros(P, 0, 3))
            gm._opacity = par(torch.randn(P, 1, generator=g) * 2.0)              # sigmoid: ~28 % below 0.3
            gm._marker = par((torch.rand(P, 1, generator=g) < 0.3).float() * torch.rand(P, 1, generator=g))
            gm._kp_score = par(torch.rand(P, 1, generator=g))
            # log-scales around log(0.06): both the clone (small) and the split (large) branch are taken;
            # a few huge ones trip the 0.1 * extent world-size prune
            ls = np.log(0.06) + 0.8 * torch.randn(P, 3, generator=g)
            ls[:6] = np.log(0.9)
            gm._scaling = par(ls)
            gm._rotation = par(torch.randn(P, 4, generator=g))                   # un-normalised, as stored
            gm.max_radii2D = torch.randint(0, 40, (P,), generator=g).float()
            args = types.SimpleNamespace(percent_dense=pd, position_lr_init=0.0016, position_lr_final=0.0000016,
                                         position_lr_delay_mult=0.01, position_lr_max_steps=30000, feature_lr=0.0025,
                                         opacity_lr=0.05, marker_lr=0.05, kp_score_lr=0.05, scaling_lr=0.001,
                                         rotation_lr=0.001)
            gm.spatial_lr_scale = 6.0
            gm.training_setup(args)
            out[case + "hyper"] = np.array([extent, pd, 0.0002, 0.3, 20.0])     # extent, percent_dense, max_grad, min_opacity, size_threshold
            snapshot(gm, case + "s0_", out)
            adam_steps(gm, g, 3, case + "a_", out, first_iteration=1)
            snapshot(gm, case + "s1_", out)
            # densification statistics: some rows never seen (denom 0 -> NaN -> 0)
            gm.xyz_gradient_accum = torch.rand(P, 1, generator=g) * 0.002
            gm.denom = torch.randint(0, 4, (P, 1), generator=g).float()
            out[case + "accum_in"] = gm.xyz_gradient_accum.numpy().copy()
            out[case + "denom_in"] = gm.denom.numpy().copy()
            out[case + "max_radii_in"] = gm.max_radii2D.numpy().copy()
            unit = torch.randn(2, P, 3, generator=g)
            out[case + "unit_noise"] = unit.numpy().copy()

            real_normal = torch.normal
            scal_before = gm.get_scaling.detach().clone()

            def injected_normal(mean, std, **kw):
                # rows of std = get_scaling[selected].repeat(2, 1): identify the source rows by value
                n = std.shape[0] // 2
                key = {tuple(r.tolist()): i for i, r in enumerate(scal_after_clone)}
                idx = torch.tensor([key[tuple(r.tolist())] for r in std[:n]], dtype=torch.long)
                assert torch.equal(std[:n], std[n:])
                src = idx                                # clones are never split (their padded grad is 0)
                assert int(src.max()) < P
                return torch.cat((unit[0, src], unit[1, src]), 0) * std + mean

            # get_scaling after the clone step = rows [0, P) unchanged + clones appended: the first match of a
            # value is always the original row (clones repeat an original's scaling; they are not selected)
            scal_after_clone = scal_before
            torch.normal = injected_normal
            gmod.torch.normal = injected_normal
            try:
                gm.densify_and_prune(0.0002, 0.3, extent, 20.0)
            finally:
                torch.normal = real_normal
                gmod.torch.normal = real_normal
            snapshot(gm, case + "s2_", out)
            adam_steps(gm, g, 2, case + "b_", out, first_iteration=4)
            snapshot(gm, case + "s3_", out)
            print(case, "P", P, "->", gm._xyz.shape[0])
    path = os.path.join(HERE, "densify.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
