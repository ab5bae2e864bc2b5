"""The activation / SH-packing oracle (oracle/activations.py) against the fixture recorded from
the reference's own render() + autograd (tests/golden/activations.npz).  CPU only."""
import os

import numpy as np
import pytest

from oracle import activations as act

GOLD = os.path.join(os.path.dirname(__file__), "golden", "activations.npz")
CASES = ("deg0", "deg2of3", "deg3")


def load_case(name):
    d = np.load(GOLD)
    return {k[len(name) + 1:]: d[k] for k in d.files if k.startswith(name + "_")}


@pytest.mark.parametrize("name", CASES)
def test_forward_matches_reference_render(name):
    c = load_case(name)
    out = act.forward(c["raw_xyz"], c["raw_f_dc"], c["raw_f_rest"], c["raw_scaling"], c["raw_rotation"],
                      c["raw_opacity"], c["raw_kp_score"], c["campos"], int(c["active_sh_degree"]))
    np.testing.assert_allclose(out["scales"], c["out_scales"], rtol=2e-7, atol=0)
    np.testing.assert_allclose(out["rotations"], c["out_rotations"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(out["opacities"], c["out_opacities"], rtol=0, atol=2e-7)   # 1 ulp: exp implementations differ
    np.testing.assert_allclose(out["colors"], c["out_colors_precomp"], rtol=0, atol=2e-6)
    assert np.array_equal(c["out_means3D"], c["raw_xyz"])          # get_xyz is the raw tensor
    assert (out["colors"][:, :3] == 0).any()                       # the clamp is exercised


@pytest.mark.parametrize("name", CASES)
def test_backward_matches_reference_autograd(name):
    c = load_case(name)
    g = act.backward(c["raw_xyz"], c["raw_f_dc"], c["raw_f_rest"], c["raw_scaling"], c["raw_rotation"],
                     c["raw_opacity"], c["raw_kp_score"], c["campos"], int(c["active_sh_degree"]),
                     c["G_scales"], c["G_rotations"], c["G_opacities"], c["G_colors_precomp"])

    def close(a, b, what):
        assert a.shape == b.shape, what
        if b.size == 0:
            return
        scale = max(np.abs(b).max(), 1e-30)
        assert np.abs(a - b).max() <= 2e-5 * scale + 1e-7, (what, np.abs(a - b).max(), scale)

    close(g["d_scaling"], c["grad_scaling"], "scaling")
    close(g["d_rotation"], c["grad_rotation"], "rotation")
    close(g["d_opacity"], c["grad_opacity"], "opacity")
    close(g["d_f_dc"], c["grad_f_dc"], "f_dc")
    close(g["d_f_rest"], c["grad_f_rest"], "f_rest")
    close(g["d_extras"], c["grad_kp_score"], "kp_score")
    # _xyz receives the rasterizer's means3D gradient unchanged plus the view-direction term
    close(g["d_xyz"] + c["G_means3D"], c["grad_xyz"], "xyz")
    if int(c["active_sh_degree"]) == 0:
        assert np.abs(g["d_xyz"]).max() == 0.0
    else:
        assert np.abs(g["d_xyz"]).max() > 0.0
    K = (int(c["active_sh_degree"]) + 1) ** 2
    assert np.all(g["d_f_rest"][:, K - 1:] == 0)                   # inactive coefficients get no gradient


def test_isotropic_scaling_is_repeated():
    c = load_case("deg0")
    iso = c["raw_scaling"][:, :1]
    out = act.forward(c["raw_xyz"], c["raw_f_dc"], c["raw_f_rest"], iso, c["raw_rotation"], c["raw_opacity"],
                      c["raw_kp_score"], c["campos"], 0)
    assert out["scales"].shape[1] == 3 and np.array_equal(out["scales"][:, 0], out["scales"][:, 2])
    g = act.backward(c["raw_xyz"], c["raw_f_dc"], c["raw_f_rest"], iso, c["raw_rotation"], c["raw_opacity"],
                     c["raw_kp_score"], c["campos"], 0, c["G_scales"], c["G_rotations"], c["G_opacities"],
                     c["G_colors_precomp"])
    np.testing.assert_allclose(g["d_scaling"][:, 0], (c["G_scales"] * out["scales"]).sum(1), rtol=1e-4, atol=1e-7)
