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
        for grp in self.param_groups:
            if betas is None:
                betas, eps = grp["betas"], grp["eps"]
            elif betas != grp["betas"] or eps != grp["eps"]:
                raise RuntimeError("splatloc_amd.optim.Adam: all groups must share betas and eps")
            for p in grp["params"]:
                if p.grad is None:
                    continue
                _require_gpu(p, "parameter")
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("splatloc_amd.optim.Adam: parameters must be contiguous float32")
                dev = p.device
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.to(torch.float32).contiguous()
                gate = None
                if self.key_gate is not None and grp.get("name") == self.key_gate[2] and p.numel():
                    gate = self.key_gate[0].detach()
                    if gate.dtype != torch.float32 or not gate.is_contiguous():
                        gate = gate.to(torch.float32).contiguous()
                    if gate.numel() != p.shape[0]:
                        raise RuntimeError("splatloc_amd.optim.Adam: key gate needs one value per parameter row")
                keep += [g, gate]
                width = int(p.numel() // p.shape[0]) if p.dim() and p.shape[0] else 1
                entries.append(_native.AdamGroup(
                    C.c_void_p(p.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(st["exp_avg"].data_ptr()),
                    C.c_void_p(st["exp_avg_sq"].data_ptr()), None if gate is None else C.c_void_p(gate.data_ptr()),
                    p.numel(), max(width, 1), float(grp["lr"]), float(st["step"])))
        if entries:
            if len(entries) > 16:
                raise RuntimeError("splatloc_amd.optim.Adam: at most 16 parameters per step")
            arr = (_native.AdamGroup * len(entries))(*entries)
            thr = self.key_gate[1] if self.key_gate is not None else 0.0
            with _on_device(dev):
                _native.check(lib.splatraster_adam_step(len(entries), arr, C.c_double(betas[0]), C.c_double(betas[1]),
                                                        C.c_double(eps), C.c_float(thr), _stream(dev)), "adam_step")
        del keep
        return loss
