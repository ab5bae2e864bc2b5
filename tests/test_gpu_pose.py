"""Pose-gradient extension on the GPU: dL/dviewmatrix, dL/dprojmatrix, dL/dcampos vs the oracle
(itself checked against fp64 autograd in tests/test_oracle_autograd.py), and a pose refinement
loop in the shape the reference's utils/optimization_utils.py helpers are meant for
(axis-angle + translation -> 4x4 transform -> viewmatrix -> rasterizer)."""
import numpy as np
import pytest
import torch

from splatloc_amd.camera import PinholeCamera
from splatloc_amd.synthetic import make_scene
from tests.helpers import assert_grad_close, oracle_backward, oracle_forward

pytestmark = pytest.mark.gpu


def _axis_angle_to_matrix(w):
    """Rodrigues, same formula as utils/optimization_utils.py:5-22 (own restatement)."""
    theta = torch.linalg.norm(w) + 1e-12
    k = w / theta
    zero = torch.zeros((), dtype=w.dtype, device=w.device)
    K = torch.stack([torch.stack([zero, -k[2], k[1]]), torch.stack([k[2], zero, -k[0]]),
                     torch.stack([-k[1], k[0], zero])])
    return torch.eye(3, dtype=w.dtype, device=w.device) + torch.sin(theta) * K + (1 - torch.cos(theta)) * (K @ K)


def _camera_tensors(cam, R, t):
    """world_view_transform / full_proj_transform / camera_center as differentiable functions of (R, t)
    (utils/camera_utils.py:129-139 semantics)."""
    dev = R.device
    Rt = torch.eye(4, device=dev)
    Rt = torch.cat([torch.cat([R, t[:, None]], dim=1), torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=dev)], dim=0)
    view = Rt.transpose(0, 1)
    proj = view @ cam.projection_matrix.to(dev)
    campos = torch.linalg.inv(view)[3, :3]
    return view, proj, campos


@pytest.mark.parametrize("use_sh", [False, True])
def test_pose_gradients_match_oracle(use_sh):
    from splatloc_amd import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda:0")
    C = 3 if use_sh else 4
    sc = make_scene(3000, 256, 192, C, 70 + int(use_sh), scale_median=0.03)
    ang = 0.15
    R = torch.tensor([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], dtype=torch.float32)
    sc.camera = PinholeCamera(256, 192, 128.0, 128.0, 127.5 + 0.4, 95.5 - 0.2, R, torch.tensor([0.05, -0.03, 0.2]))
    g = torch.Generator().manual_seed(3)
    shs = 0.5 * torch.randn(3000, 16, 3, generator=g) if use_sh else None
    kw = dict(sh_degree=2, colors_precomp=None, shs=shs.numpy()) if use_sh else {}
    f = oracle_forward(sc, **kw)
    b = oracle_backward(f, sc)
    cam = sc.camera
    view = cam.world_view_transform.to(dev).clone().requires_grad_(True)
    proj = cam.full_proj_transform.to(dev).clone().requires_grad_(True)
    campos = cam.camera_center.to(dev).clone().requires_grad_(True)
    rs = GaussianRasterizationSettings(192, 256, cam.tanfovx, cam.tanfovy, sc.bg.to(dev), 1.0, view, proj,
                                       2 if use_sh else 0, campos, False, False)
    t = lambda x: x.to(dev)  # noqa: E731
    color, depth, alpha, radii = GaussianRasterizer(raster_settings=rs)(
        means3D=t(sc.means3D), means2D=torch.zeros(3000, 3, device=dev), shs=t(shs) if use_sh else None,
        colors_precomp=None if use_sh else t(sc.features), opacities=t(sc.opacities), scales=t(sc.scales),
        rotations=t(sc.rotations), cov3D_precomp=None)
    ((color * t(sc.dL_dcolor)).sum() + (depth * t(sc.dL_ddepth)).sum() + (alpha * t(sc.dL_dalpha)).sum()).backward()
    assert_grad_close("dL_dviewmatrix", view.grad.cpu().numpy(), b["dL_dviewmatrix"], rtol=3e-3, atol_scale=3e-4)
    assert_grad_close("dL_dprojmatrix", proj.grad.cpu().numpy(), b["dL_dprojmatrix"], rtol=3e-3, atol_scale=3e-4)
    if use_sh:
        assert_grad_close("dL_dcampos", campos.grad.cpu().numpy(), b["dL_dcampos"], rtol=3e-3, atol_scale=3e-4)
        assert float(np.abs(b["dL_dcampos"]).max()) > 0
    else:
        assert campos.grad is None or float(campos.grad.abs().max()) == 0.0


def test_pose_refinement_converges():
    """Recover a perturbed camera pose by gradient descent on a photometric + depth loss."""
    from splatloc_amd import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda:0")
    sc = make_scene(6000, 256, 192, 3, 80, scale_median=0.05).to(dev)
    cam = PinholeCamera(256, 192, 128.0, 128.0, 127.5, 95.5)

    def render(w, tr):
        view, proj, campos = _camera_tensors(cam, _axis_angle_to_matrix(w), tr)
        rs = GaussianRasterizationSettings(192, 256, cam.tanfovx, cam.tanfovy, sc.bg, 1.0, view, proj, 0, campos,
                                           False, False)
        return GaussianRasterizer(raster_settings=rs)(
            means3D=sc.means3D, means2D=torch.zeros_like(sc.means3D), shs=None, colors_precomp=sc.features,
            opacities=sc.opacities, scales=sc.scales, rotations=sc.rotations, cov3D_precomp=None)

    w_true = torch.tensor([0.02, -0.03, 0.01], device=dev)
    t_true = torch.tensor([0.03, -0.02, 0.05], device=dev)
    with torch.no_grad():
        tgt_c, tgt_d, _, _ = render(w_true, t_true)
    w = torch.tensor([1e-4, 1e-4, 1e-4], device=dev, requires_grad=True)
    tr = torch.zeros(3, device=dev, requires_grad=True)
    opt = torch.optim.Adam([{"params": [w], "lr": 2e-3}, {"params": [tr], "lr": 3e-3}])
    err0 = float((w.detach() - w_true).norm() + (tr.detach() - t_true).norm())
    first = None
    for it in range(150):
        color, depth, alpha, _ = render(w, tr)
        loss = (color - tgt_c).abs().mean() + 0.2 * (depth - tgt_d).abs().mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        assert torch.isfinite(w.grad).all() and torch.isfinite(tr.grad).all()
        opt.step()
        first = float(loss) if first is None else first
    err1 = float((w.detach() - w_true).norm() + (tr.detach() - t_true).norm())
    assert float(loss) < 0.35 * first, (first, float(loss))
    assert err1 < 0.35 * err0, (err0, err1)


def test_refine_pose_recovers_a_perturbed_camera():
    """splatloc_amd.pose.refine_pose — the loop the reference's utils/optimization_utils.py helpers are meant for (axis-angle +
    translation -> 4x4 -> viewmatrix -> rasterizer -> photometric + depth loss -> pose gradients): a camera perturbed by
    ~2 degrees / 6 cm ends within 30 % of its initial pose error, with the loss down by 3x."""
    from splatloc_amd import GaussianRasterizationSettings, GaussianRasterizer, pose
    dev = torch.device("cuda:0")
    sc = make_scene(6000, 256, 192, 3, 81, scale_median=0.05).to(dev)
    cam = PinholeCamera(256, 192, 128.0, 128.0, 127.5, 95.5)
    cam.to(dev)
    w_true = torch.tensor([[0.02, -0.03, 0.01]], device=dev)
    t_true = torch.tensor([[0.03, -0.02, 0.05]], device=dev)
    W2C_true = pose.at_to_transform_matrix(w_true, t_true)[0]
    with torch.no_grad():
        view, proj, campos = pose.camera_tensors(W2C_true, cam.projection_matrix)
        rs = GaussianRasterizationSettings(192, 256, cam.tanfovx, cam.tanfovy, sc.bg, 1.0, view, proj, 0, campos, False, False)
        tgt_c, tgt_d, _, _ = GaussianRasterizer(raster_settings=rs)(
            means3D=sc.means3D, means2D=torch.zeros_like(sc.means3D), shs=None, colors_precomp=sc.features,
            opacities=sc.opacities, scales=sc.scales, rotations=sc.rotations, cov3D_precomp=None)
    g = dict(means3D=sc.means3D, colors=sc.features, opacities=sc.opacities, scales=sc.scales, rotations=sc.rotations)
    W2C, hist = pose.refine_pose((tgt_c, tgt_d), g, cam, torch.eye(4, device=dev), iterations=150, background=sc.bg)
    hist = hist.cpu()
    assert torch.isfinite(hist).all() and float(hist[-1]) < 0.35 * float(hist[0]), (float(hist[0]), float(hist[-1]))

    def err(M):
        dR = M[:3, :3] @ W2C_true[:3, :3].T
        ang = torch.acos(((torch.trace(dR) - 1) / 2).clamp(-1, 1))
        return float(ang) + float((M[:3, 3] - W2C_true[:3, 3]).norm())
    assert err(W2C) < 0.35 * err(torch.eye(4, device=dev)), (err(torch.eye(4, device=dev)), err(W2C))


def test_graph_free_refine_pose_follows_the_autograd_loop():
    """The graph-free loop (csrc/pose.hip: fused L1 loss, chain rule + Adam + next camera tensors in one single-thread kernel)
    against round 4's loop (autograd through the 4x4 algebra, torch.optim.Adam): same loss history and the same pose to float32
    rounding over 40 iterations — a check of the hand-written chain rule (dL/dview, dL/dproj, dL/dcampos -> axis-angle,
    translation) and of the in-kernel Adam."""
    from splatloc_amd import GaussianRasterizationSettings, GaussianRasterizer, pose
    dev = torch.device("cuda:0")
    sc = make_scene(6000, 256, 192, 3, 81, scale_median=0.05).to(dev)
    cam = PinholeCamera(256, 192, 128.0, 128.0, 127.5 + 0.3, 95.5 - 0.2)
    cam.to(dev)
    W2C_true = pose.at_to_transform_matrix(torch.tensor([[0.02, -0.03, 0.01]], device=dev),
                                           torch.tensor([[0.03, -0.02, 0.05]], device=dev))[0]
    W2C0 = pose.at_to_transform_matrix(torch.tensor([[0.01, 0.02, -0.015]], device=dev),
                                       torch.tensor([[-0.02, 0.01, 0.03]], device=dev))[0]        # a non-trivial start frame
    with torch.no_grad():
        view, proj, campos = pose.camera_tensors(W2C_true, cam.projection_matrix)
        rs = GaussianRasterizationSettings(192, 256, cam.tanfovx, cam.tanfovy, sc.bg, 1.0, view, proj, 0, campos, False, False)
        tgt_c, tgt_d, _, _ = GaussianRasterizer(raster_settings=rs)(
            means3D=sc.means3D, means2D=torch.zeros_like(sc.means3D), shs=None, colors_precomp=sc.features,
            opacities=sc.opacities, scales=sc.scales, rotations=sc.rotations, cov3D_precomp=None)
    g = dict(means3D=sc.means3D, colors=sc.features, opacities=sc.opacities, scales=sc.scales, rotations=sc.rotations)
    Wa, ha = pose.refine_pose((tgt_c, tgt_d), g, cam, W2C0, iterations=40, background=sc.bg, graph_free=False)
    Wg, hg = pose.refine_pose((tgt_c, tgt_d), g, cam, W2C0, iterations=40, background=sc.bg, graph_free=True)
    ha, hg = ha.cpu().double(), hg.cpu().double()
    assert torch.isfinite(hg).all()
    # iteration 0 evaluates the same pose: equal to summation-order rounding; afterwards the trajectories stay together
    assert abs(float(ha[0] - hg[0])) <= 2e-6 * float(ha[0])
    assert float((ha - hg).abs().max()) <= 2e-3 * float(ha[0]), (ha[-5:], hg[-5:])
    assert float((Wa - Wg).abs().max()) <= 2e-4, (Wa, Wg)
    assert float(hg[-1]) < 0.7 * float(hg[0])


def test_pose_step_kernel_matches_autograd_and_torch_adam():
    """splatraster_pose_step alone: random upstream gradients for (viewmatrix, projmatrix, campos), three steps — the six
    parameters, the Adam moments and the camera tensors it writes against torch.autograd + torch.optim.Adam in float64."""
    import ctypes as C
    from splatloc_amd import _native, pose
    lib = _native.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    W2C0 = pose.at_to_transform_matrix(0.3 * torch.randn(1, 3, generator=g), torch.randn(1, 3, generator=g))[0].to(dev)
    Pm = torch.randn(4, 4, generator=g).to(dev)
    state = torch.zeros(20, device=dev)
    view, proj, campos = torch.empty(4, 4, device=dev), torch.empty(4, 4, device=dev), torch.empty(3, device=dev)
    ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
    w = torch.zeros(1, 3, dtype=torch.float64, requires_grad=True)
    t = torch.zeros(1, 3, dtype=torch.float64, requires_grad=True)
    opt = torch.optim.Adam([{"params": [w], "lr": 2e-2}, {"params": [t], "lr": 3e-2}])
    _native.check(lib.splatraster_pose_step(None, None, None, ptr(W2C0), ptr(Pm), 2e-2, 3e-2, 0.9, 0.999, 1e-8, 0, ptr(state),
                                            ptr(view), ptr(proj), ptr(campos), None), "pose_step")
    for k in range(3):
        Gv, Gp, Gc = (torch.randn(4, 4, generator=g).to(dev), torch.randn(4, 4, generator=g).to(dev),
                      torch.randn(3, generator=g).to(dev))
        W2C = pose.at_to_transform_matrix(w, t)[0] @ W2C0.double().cpu()
        v_ref, p_ref, c_ref = pose.camera_tensors(W2C, Pm.double().cpu())
        torch.cuda.synchronize()
        assert torch.allclose(view.cpu().double(), v_ref.detach(), atol=2e-6) and torch.allclose(proj.cpu().double(), p_ref.detach(), atol=1e-5)
        assert torch.allclose(campos.cpu().double(), c_ref.detach(), atol=1e-5)
        loss = (v_ref * Gv.cpu().double()).sum() + (p_ref * Gp.cpu().double()).sum() + (c_ref * Gc.cpu().double()).sum()
        opt.zero_grad()
        loss.backward()
        opt.step()
        _native.check(lib.splatraster_pose_step(ptr(Gv), ptr(Gp), ptr(Gc), ptr(W2C0), ptr(Pm), 2e-2, 3e-2, 0.9, 0.999, 1e-8, 1,
                                                ptr(state), ptr(view), ptr(proj), ptr(campos), None), "pose_step")
        torch.cuda.synchronize()
        st = state.cpu().double()
        assert torch.allclose(st[:3], w.detach()[0], atol=2e-6), (k, st[:3], w)
        assert torch.allclose(st[3:6], t.detach()[0], atol=2e-6), (k, st[3:6], t)
        assert float(st[18]) == k + 1


@pytest.mark.parametrize("name,C", [("S2-ref-layout", 4), ("S2-640", 35)])
def test_pose_gradients_full_size_scenes12_intrinsics(name, C):
    """BASELINE config 4's shape: 500k Gaussians, 640x480 at the 12-Scenes intrinsics (fx = fy = 572, cx = 320, cy = 240:
    configs/scenes12/base_config.yaml:17-27), SplatLoc's [rgb | kp] layout (C = 4) and the north-star channel count (C = 35),
    a rotated and translated camera: dL/dviewmatrix, dL/dprojmatrix against the oracle (double accumulation), every element
    within 1e-3 of its own magnitude + 1e-4 of the tensor's scale (round 4 checked 3 000 Gaussians at 256x192 to 3e-3)."""
    from splatloc_amd import GaussianRasterizationSettings, GaussianRasterizer
    from splatloc_amd.synthetic import WORKLOADS
    dev = torch.device("cuda:0")
    wl = dict(WORKLOADS[name])
    sc = make_scene(**wl)
    ang = 0.06
    R = torch.tensor([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], dtype=torch.float32)
    sc.camera = PinholeCamera(640, 480, 572.0, 572.0, 320.0, 240.0, R, torch.tensor([0.04, -0.02, 0.1]))
    f = oracle_forward(sc)
    b = oracle_backward(f, sc)
    cam = sc.camera
    view = cam.world_view_transform.to(dev).clone().requires_grad_(True)
    proj = cam.full_proj_transform.to(dev).clone().requires_grad_(True)
    campos = cam.camera_center.to(dev).clone().requires_grad_(True)
    rs = GaussianRasterizationSettings(480, 640, cam.tanfovx, cam.tanfovy, sc.bg.to(dev), 1.0, view, proj, 0, campos, False, False)
    t = lambda x: x.to(dev)  # noqa: E731
    P = sc.means3D.shape[0]
    color, depth, alpha, radii = GaussianRasterizer(raster_settings=rs)(
        means3D=t(sc.means3D), means2D=torch.zeros(P, 3, device=dev), shs=None, colors_precomp=t(sc.features),
        opacities=t(sc.opacities), scales=t(sc.scales), rotations=t(sc.rotations), cov3D_precomp=None)
    assert int(color.grad_fn.num_rendered) == f["num_rendered"]
    ((color * t(sc.dL_dcolor)).sum() + (depth * t(sc.dL_ddepth)).sum() + (alpha * t(sc.dL_dalpha)).sum()).backward()
    assert float(np.abs(b["dL_dviewmatrix"]).max()) > 0
    assert_grad_close("dL_dviewmatrix", view.grad.cpu().numpy(), b["dL_dviewmatrix"], rtol=1e-3, atol_scale=1e-4)
    assert_grad_close("dL_dprojmatrix", proj.grad.cpu().numpy(), b["dL_dprojmatrix"], rtol=1e-3, atol_scale=1e-4)
