"""Adam over all parameter groups of the reference's GaussianModel in ONE HIP launch.

`Adam` subclasses `torch.optim.Adam` and keeps its state layout (`state[p] = {step, exp_avg,
exp_avg_sq}`), so it is a drop-in at gaussian_model.py:287
(`self.optimizer = torch.optim.Adam(l, lr=0.0, eps=1e-15)`): the reference's own optimizer surgery
(`cat_tensors_to_optimizer`, `_prune_optimizer`, `replace_tensor_to_optimizer`) and
`splatloc_amd.densify.densify_and_prune` both keep working on it.  `step()` issues one kernel for
every group that has a gradient (a parameter whose `.grad is None` — SplatLoc's marker — is skipped
and gets no state, as in torch).  `key_gate=(tensor [P,1], threshold)` reproduces the key-primitive
freeze `gaussians.get_xyz.grad[key_mask] = 0` (train_gaussians.py:231-234) inside the same launch,
without the reference's `.cpu()` synchronisation.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _native
from .rasterizer import _require_gpu, _stream, _on_device


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False)
        self.key_gate = None        # (gate tensor with one value per xyz row, threshold, group name)
        self._radii_update = None   # (radii, max_radii2D) for the next step() — set_radii_update

    def set_key_gate(self, gate: torch.Tensor, threshold: float = 0.005, group: str = "xyz") -> None:
        """Rows of `group` whose gate value exceeds `threshold` see a zero gradient in step()."""
        self.key_gate = None if gate is None else (gate, float(threshold), group)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _native.load()
        entries, keep = [], []
        betas = eps = None
        dev = None
        f32 = torch.float32
        gate_spec = self.key_gate
        for grp in self.param_groups:
            if betas is None:
                betas, eps = grp["betas"], grp["eps"]
            elif betas != grp["betas"] or eps != grp["eps"]:
                raise RuntimeError("splatloc_amd.optim.Adam: all groups must share betas and eps")
            for p in grp["params"]:
                g = p.grad
                if g is None:
                    continue
                if not p.is_cuda:
                    _require_gpu(p, "parameter")
                if p.dtype is not f32 or not p.is_contiguous():
                    raise RuntimeError("splatloc_amd.optim.Adam: parameters must be contiguous float32")
                dev = p.device
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                # `step` stays a 0-dim CPU tensor (torch.optim.Adam's state layout: the reference's optimizer surgery carries it
                # over); it is advanced through its numpy view — an in-place tensor op costs ~4 us per group, seven per step
                stp = st["step"]
                if stp.is_cuda:
                    stp += 1
                    step_val = float(stp)
                else:
                    view = stp.numpy()
                    view += 1
                    step_val = float(view)
                if g.dtype is not f32 or not g.is_contiguous():
                    g = g.to(f32).contiguous()
                gate = None
                if gate_spec is not None and grp.get("name") == gate_spec[2] and p.numel():
                    gate = gate_spec[0].detach()
                    if gate.dtype is not f32 or not gate.is_contiguous():
                        gate = gate.to(f32).contiguous()
                    if gate.numel() != p.shape[0]:
                        raise RuntimeError("splatloc_amd.optim.Adam: key gate needs one value per parameter row")
                keep.append(g)
                keep.append(gate)
                n = p.numel()
                rows = p.shape[0] if p.dim() else 0
                width = (n // rows) if rows else 1
                entries.append(_native.AdamGroup(
                    p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                    None if gate is None else gate.data_ptr(), n, max(width, 1), float(grp["lr"]), step_val))
        radii_update, self._radii_update = getattr(self, "_radii_update", None), None
        if len(entries) > 16:
            raise RuntimeError("splatloc_amd.optim.Adam: at most 16 parameters per step")
        if entries or radii_update is not None:
            arr = (_native.AdamGroup * max(len(entries), 1))(*entries)
            thr = gate_spec[1] if gate_spec is not None else 0.0
            if betas is None:
                betas, eps = (0.9, 0.999), 1e-8
            if radii_update is not None:      # the frame's max_radii2D line rides this launch (set_radii_update)
                radii, max_radii = radii_update
                dev = radii.device if dev is None else dev
                with _on_device(dev):
                    _native.check(lib.splatraster_adam_step_radii(len(entries), arr, C.c_double(betas[0]), C.c_double(betas[1]),
                                                                  C.c_double(eps), C.c_float(thr), int(radii.numel()), radii.data_ptr(),
                                                                  max_radii.data_ptr(), _stream(dev)), "adam_step_radii")
            else:
                with _on_device(dev):
                    _native.check(lib.splatraster_adam_step(len(entries), arr, C.c_double(betas[0]), C.c_double(betas[1]),
                                                            C.c_double(eps), C.c_float(thr), _stream(dev)), "adam_step")
        del keep
        return loss

    def set_radii_update(self, radii, max_radii2D):
        """The NEXT `step()` also performs `max_radii2D[vis] = max(max_radii2D[vis], radii[vis])` (vis = radii > 0), the statistics
        line of SplatLoc.color_refinement (train_gaussians.py:293-294), inside its one launch."""
        if radii is None:
            self._radii_update = None
            return
        if radii.dtype != torch.int32 or not radii.is_contiguous() or not radii.is_cuda:
            raise RuntimeError("set_radii_update: `radii` must be the contiguous int32 device tensor of a forward")
        if (max_radii2D.dtype != torch.float32 or not max_radii2D.is_contiguous() or max_radii2D.numel() != radii.numel()
                or max_radii2D.device != radii.device):
            raise RuntimeError("set_radii_update: `max_radii2D` must be a contiguous float32 tensor with one element per Gaussian")
        self._radii_update = (radii, max_radii2D)
