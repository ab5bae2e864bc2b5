"""Frame-parallel data parallelism for SplatLoc's mapping step (one process per GPU).

In `SplatLoc.map` one optimisation step already sums the loss over `window_size` = 5
independent views before a single backward / optimizer step (train_gaussians.py:195-229,
265), so dealing the views of the window to ranks and SUM-all-reducing the parameter
gradients is mathematically the reference step.  Every rank holds a full
replica of the Gaussian scene; there is no collective on the data path of a single frame.

Collectives (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests) — `reduce_step`, at most
TWO per optimisation step:
  * ONE SUM over [parameter gradients (xyz, features, opacity, scaling, rotation, kp_score) |
         densification statistics increments (xyz_gradient_accum, denom; gaussian_model.py:677-679)]
  * ONE MAX over [max_radii2D (train_gaussians.py:240-244) | visibility flags of an opacity-reset step
         (gaussian_model.py:384-392)]
xGMI is point-to-point (7 links/GPU): the payload (64 B/Gaussian at the reference layout,
+4 B per extra feature channel) is sent as one large buffer, never per-parameter-row.
`allreduce_grads` / `sync_densification_stats` are the round-1..3 building blocks (kept: three collectives).
"""
from __future__ import annotations

import os
import time
from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist

# A process group of ONE rank normally exchanges nothing (every function below returns early).  `force=True` (or
# SPLATLOC_FORCE_COLLECTIVES=1 for code that cannot pass the argument) issues every collective anyway: a world-size-1 RCCL
# group on the one GPU a builder box has exercises the exact calls the 8-GPU run makes — the in-place span SUM on the
# backward's own allocation, the MAX, the aliased reduce-scatter + all-gather pair, the header, the broadcast
# (tests/test_gpu_rccl.py, bench.py --force-process-group).  A SUM / MAX / broadcast over one rank leaves the values unchanged.
FORCE_COLLECTIVES = os.environ.get("SPLATLOC_FORCE_COLLECTIVES", "0") == "1"


def collectives_active(group=None, force: bool = False) -> bool:
    """True when the functions of this module will issue collectives: a process group exists and has more than one rank
    (or `force` / FORCE_COLLECTIVES asks for them on a group of one)."""
    if not dist.is_available() or not dist.is_initialized():
        return False
    return dist.get_world_size(group) > 1 or force or FORCE_COLLECTIVES


def shard_views(view_ids: Sequence[int], rank: int, world_size: int) -> List[int]:
    """Round-robin deal of the window's views to ranks (rank r gets views r, r+W, ...)."""
    return [v for i, v in enumerate(view_ids) if i % world_size == rank]


def _shared_spans(ts: List[torch.Tensor]):
    """Find tensors that are dense views carved out of ONE allocation (the rasterizer's
    backward returns all parameter gradients of a frame that way) and return
    (spans, rest): `spans` are flat tensors aliasing [lowest, highest) of each such
    allocation — reducing a span reduces every member in place, with no staging copy —
    and `rest` are the tensors that are not part of any span.

    Members must be contiguous, non-overlapping, and separated by less than 16 bytes
    (alignment padding): anything else in between is not ours to sum.
    """
    groups = {}
    for t in ts:
        key = (t.untyped_storage().data_ptr(), t.dtype, t.device)
        groups.setdefault(key, []).append(t)
    spans, rest = [], []
    for (_, dtype, device), members in groups.items():
        if len(members) < 2 or not all(m.is_contiguous() and m.numel() for m in members):
            rest.extend(members)
            continue
        members = sorted(members, key=lambda m: m.storage_offset())
        esize = members[0].element_size()
        ok = True
        for a, b in zip(members[:-1], members[1:]):
            gap = b.storage_offset() - (a.storage_offset() + a.numel())
            if gap < 0 or gap * esize >= 16:
                ok = False
                break
        if not ok:
            rest.extend(members)
            continue
        lo = members[0].storage_offset()
        hi = members[-1].storage_offset() + members[-1].numel()
        span = torch.empty(0, dtype=dtype, device=device).set_(members[0].untyped_storage(), lo, (hi - lo,), (1,))
        spans.append(span)
    return spans, rest


def allreduce_grads(tensors: Iterable[Optional[torch.Tensor]], group=None, bucket_bytes: int = 256 << 20, force: bool = False):
    """SUM-all-reduce gradient tensors in place, as few large buffers.

    Gradients that already live side by side in one allocation (see _shared_spans) are
    reduced where they are, as one call; the rest are coalesced into flat buckets.
    Tensors that are None are skipped (a rank whose views saw no Gaussian still has dense
    zero grads from the rasterizer, so shapes agree across ranks).

    Returns which path the payload took, e.g. {"spans": 1, "span_bytes": 92000000, "buckets": 0,
    "bucket_bytes": 0} — the in-place span path depends on autograd keeping the rasterizer's gradient
    views as `.grad` (it does when a parameter's first gradient of the step is one of them); a silent
    fall-back to the staged bucket path would otherwise go unnoticed (None when not distributed).
    """
    if not collectives_active(group, force):
        return None
    with torch.no_grad():
        spans, ts = _shared_spans([t for t in tensors if t is not None])
    bucket, size = [], 0
    pending = []
    path = {"spans": len(spans), "span_bytes": sum(sp.numel() * sp.element_size() for sp in spans), "buckets": 0,
            "bucket_bytes": sum(t.numel() * t.element_size() for t in ts)}
    for span in spans:
        pending.append((dist.all_reduce(span, op=dist.ReduceOp.SUM, group=group, async_op=True), None, None))

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        path["buckets"] += 1
        if len(bucket) == 1 and bucket[0].is_contiguous():
            pending.append((dist.all_reduce(bucket[0], op=dist.ReduceOp.SUM, group=group, async_op=True), None, None))
        else:
            flat = torch.cat([t.reshape(-1) for t in bucket])
            pending.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True), flat, bucket))
        bucket, size = [], 0

    for t in ts:
        nbytes = t.numel() * t.element_size()
        if size and size + nbytes > bucket_bytes:
            flush()
        bucket.append(t)
        size += nbytes
    flush()
    for work, flat, parts in pending:
        work.wait()
        if flat is not None:
            off = 0
            for t in parts:
                n = t.numel()
                t.copy_(flat[off:off + n].view_as(t))
                off += n
    return path


def sync_densification_stats(grad_accum_inc: torch.Tensor, denom_inc: torch.Tensor, max_radii2D: torch.Tensor,
                             group=None, force: bool = False) -> None:
    """Make every replica densify identically: SUM the per-step increments of
    xyz_gradient_accum / denom and MAX max_radii2D (in place)."""
    if not collectives_active(group, force):
        return
    packed = torch.cat([grad_accum_inc.reshape(-1), denom_inc.reshape(-1)])
    w1 = dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group, async_op=True)
    w2 = dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX, group=group, async_op=True)
    w1.wait()
    w2.wait()
    n = grad_accum_inc.numel()
    grad_accum_inc.copy_(packed[:n].view_as(grad_accum_inc))
    denom_inc.copy_(packed[n:].view_as(denom_inc))


_HEADER_GROUPS: dict = {}


def _header_group(group):
    """The group the 32-byte length check runs on: a GLOO twin of `group` (same ranks), created once — collectively, by the first
    reduce_step / broadcast_model every rank reaches — when `group` itself is not a CPU backend.  The check's result is read on the
    HOST; on the payload's own (RCCL) group that read drained the stream in front of every step's payload: 0.3 ms of a 6.8-ms S2
    step on a group of one (bench.py --force-process-group: 700 instead of 733 frames/s), and the host could no longer run ahead of
    the GPU across the reduction (round-5 advisor).  Over gloo the host blocks for the exchange only, the GPU keeps working."""
    import weakref
    pg = group if group is not None else dist.distributed_c10d._get_default_group()
    hit = _HEADER_GROUPS.get(id(pg))
    if hit is not None and hit[0]() is pg:       # (the same group OBJECT: a re-initialised default group is a new one)
        return hit[1]
    if _backend_of(group) == "gloo":
        g = (group, True)
    else:
        try:
            g = (dist.new_group(ranks=dist.get_process_group_ranks(pg), backend="gloo"), True)
        except Exception:  # noqa: BLE001  (no gloo in this build: the check stays on the payload's group, with its device read)
            g = (group, False)
    try:
        _HEADER_GROUPS[id(pg)] = (weakref.ref(pg), g)
    except TypeError:    # (a group object that cannot be weakly referenced: kept alive by the cache instead)
        _HEADER_GROUPS[id(pg)] = ((lambda _pg=pg: _pg), g)
    return g


def prepare(group=None) -> None:
    """Create the length check's gloo twin of `group` now (a collective: every rank of `group` must call it) instead of inside the
    first reduce_step / broadcast_model — e.g. right after init_process_group, outside any timed region."""
    if dist.is_available() and dist.is_initialized():
        _header_group(group)


def _check_same_layout(numel, device, group) -> None:
    """A SUM over buffers of different lengths is undefined behaviour in RCCL (and silently wrong sums when the
    lengths agree but the piece order does not).  EVERY call verifies, with one fixed-size 32-byte MAX collective
    issued before the payload, that all ranks are about to reduce the same length.  (Rounds 3-4 cached the lengths a
    rank had already checked: a rank that had cached a length skipped the collective while a diverging peer issued
    it, pairing the peer's int64 MAX with this rank's float SUM — a hang instead of the intended error.  The check is
    only a check if every rank always takes part in it.)  Round 6: the exchange runs on a CPU (gloo) twin of the group
    (`_header_group`): no device synchronisation in front of the payload."""
    lens = [int(numel), 0] if not isinstance(numel, (tuple, list)) else [int(v) for v in numel][:2] + [0] * (2 - len(numel))
    hgroup, on_cpu = _header_group(group)
    t = torch.tensor([lens[0], -lens[0], lens[1], -lens[1]], dtype=torch.int64, device="cpu" if on_cpu else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=hgroup)
    v = t.tolist()
    if v[0] != -v[1] or v[2] != -v[3]:
        raise RuntimeError(f"frame_parallel.reduce_step: ranks disagree on the reduce buffers (SUM {-v[1]} .. {v[0]}, "
                           f"MAX {-v[3]} .. {v[2]} elements); every rank must hold the same replica and use the same gradient layout")


def _backend_of(group) -> str:
    try:
        return str(dist.get_backend(group))
    except Exception:  # noqa: BLE001
        return "?"


def _sum_all_ranks(buf: torch.Tensor, group, mode: str):
    """SUM `buf` (flat, contiguous) over the ranks, in place.  mode "ring": one all-reduce.  mode "rs_ag": reduce-scatter +
    all-gather on the same buffer (every rank owns 1/N of it between the two) — on a full xGMI mesh the direct algorithms
    use all seven links of a GPU at once where a ring is bound by one link pair; RCCL picks its algorithm
    itself, so which form wins is a measurement, and one hardware run can print both (`bench.py --reduce rs_ag`).  Needs the
    length to be a multiple of the world size: the caller pads.  gloo has no reduce-scatter: there the pair is emulated by
    the all-reduce (the plumbing is what the CPU tests cover)."""
    if mode == "rs_ag" and _backend_of(group) != "gloo":
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        n = buf.numel() // world
        mine = buf[rank * n:(rank + 1) * n]
        dist.reduce_scatter_tensor(mine, buf, op=dist.ReduceOp.SUM, group=group)
        return dist.all_gather_into_tensor(buf, mine, group=group, async_op=True)
    return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=True)


def reduce_step(grads: Sequence[torch.Tensor], sum_extras: Sequence[torch.Tensor] = (),
                max_extras: Sequence[torch.Tensor] = (), group=None, mode: str = "ring", force: bool = False):
    """Everything one frame-parallel optimisation step exchanges, in at most TWO collectives:

      (1) ONE SUM all-reduce over [grads | sum_extras] — the parameter gradients and the increments of
          xyz_gradient_accum / denom;
      (2) ONE MAX all-reduce over [max_extras] — max_radii2D and, on an opacity-reset step, the visibility flags
          (float32 tensors; updated in place).

    (1) runs IN PLACE, with no staging copy, when `grads` and `sum_extras` are adjacent pieces of one allocation — the
    layout `rasterize_window(..., grad_span=[])` produces: the window's summed parameter gradients followed by a
    [2, P] tail the statistics kernel writes its increments into.  Anything else (gradients that went through further
    autograd nodes, e.g. the fused activations of `training.map_step`) is packed by ONE `torch.cat`, reduced, and handed
    back as views of the reduced buffer (no copy back).  Every rank must pass the same shapes in the same order; the
    buffer length is verified across ranks the first time it is seen.

    `mode`: "ring" = one all-reduce for (1); "rs_ag" = reduce-scatter + all-gather on the same buffer (`_sum_all_ranks`;
    an in-place span whose length is not a multiple of the world size is packed so that it can be padded).  Before the
    payload every rank takes part in one 32-byte length check (`_check_same_layout`; info["header_collectives"]).

    Returns (grads_out, sum_extras_out, info): tensors holding the reduced values (the inputs themselves on the in-place
    path) and info = {"collectives", "sum_path": "in-place span" | "packed", "sum_bytes", "max_bytes", "mode"}.
    Single process / no process group: returns the inputs unchanged, info["collectives"] = 0 (`force=True`: a group of
    one rank issues every collective anyway — first contact with RCCL on one GPU).  info["header_ms"] is what the length
    check cost this rank: it reads its result on the host, so the stream is drained once per step in front of the payload."""
    grads, sum_extras, max_extras = list(grads), list(sum_extras), list(max_extras)
    if not collectives_active(group, force):
        return grads, sum_extras, {"collectives": 0, "sum_path": None, "sum_bytes": 0, "max_bytes": 0}
    if mode not in ("ring", "rs_ag"):
        raise ValueError(f"reduce_step: mode {mode!r}")
    info = {"collectives": 0, "header_collectives": 0, "sum_path": None, "sum_bytes": 0, "max_bytes": 0, "mode": mode}
    world = dist.get_world_size(group)
    pending = []
    members = [t for t in grads + sum_extras if t.numel()]
    packed = None
    if members:
        with torch.no_grad():
            spans, rest = _shared_spans(members) if len(members) > 1 else ([], members)
            total = sum(t.numel() for t in members)
            divisible = mode != "rs_ag" or total % world == 0
            if len(spans) == 1 and not rest and (mode != "rs_ag" or spans[0].numel() % world == 0):
                buf = spans[0]
                info["sum_path"] = "in-place span"
            elif len(members) == 1 and members[0].is_contiguous() and divisible:
                buf = members[0].view(-1)
                info["sum_path"] = "in-place span"
            else:
                parts = [t.reshape(-1) for t in members]
                if not divisible:
                    parts.append(torch.zeros(world - total % world, dtype=members[0].dtype, device=members[0].device))
                buf = packed = torch.cat(parts)
                info["sum_path"] = "packed"
        max_len = sum(t.numel() for t in max_extras)
        t_h = time.perf_counter()
        _check_same_layout((buf.numel(), max_len), buf.device, group)   # both payload lengths in one header
        info["header_collectives"] = 1
        info["header_ms"] = round(1e3 * (time.perf_counter() - t_h), 4)
        info["sum_bytes"] = buf.numel() * buf.element_size()
        pending.append(_sum_all_ranks(buf, group, mode))
        info["collectives"] += 2 if (mode == "rs_ag" and _backend_of(group) != "gloo") else 1
    mx = [t for t in max_extras if t.numel()]
    mbuf = None
    if mx:
        with torch.no_grad():
            mbuf = mx[0].view(-1) if (len(mx) == 1 and mx[0].is_contiguous()) else torch.cat([t.reshape(-1) for t in mx])
        if not members:
            t_h = time.perf_counter()
            _check_same_layout((0, mbuf.numel()), mbuf.device, group)
            info["header_collectives"] = 1
            info["header_ms"] = round(1e3 * (time.perf_counter() - t_h), 4)
        info["max_bytes"] = mbuf.numel() * mbuf.element_size()
        pending.append(dist.all_reduce(mbuf, op=dist.ReduceOp.MAX, group=group, async_op=True))
        info["collectives"] += 1
    for w in pending:
        w.wait()
    with torch.no_grad():
        if mbuf is not None and not (len(mx) == 1 and mx[0].is_contiguous()):
            off = 0
            for t in mx:
                n = t.numel()
                t.copy_(mbuf[off:off + n].view_as(t))
                off += n
        if packed is None:
            return grads, sum_extras, info
        outs, off = [], 0
        for t in grads + sum_extras:
            n = t.numel()
            outs.append(packed[off:off + n].view(t.shape) if n else t)
            off += n
    return outs[:len(grads)], outs[len(grads):], info


def broadcast_model(gaussians, src: int = 0, group=None, force: bool = False) -> int:
    """Re-establish bit-identical replicas from rank `src`: every parameter, Adam moment and step counter, the xyz learning
    rate and the densification statistics, packed into ONE broadcast.  Used after a phase that every rank ran redundantly on
    its own replica — SplatLoc.color_refinement is one view per step (train_gaussians.py:269-297): it does not shard, and
    float-atomic rounding makes redundant replicas drift apart in the last bits.  Returns the bytes broadcast (0 when not
    distributed)."""
    if not collectives_active(group, force):
        return 0
    opt = gaussians.optimizer
    tensors, scalars = [], []
    for grp in opt.param_groups:
        p = grp["params"][0]
        tensors.append(p.data)
        st = opt.state.get(p, None)
        if st and "exp_avg" in st:
            tensors += [st["exp_avg"], st["exp_avg_sq"]]
            scalars.append(("step", st))
        scalars.append(("lr", grp))
    tensors += [gaussians.xyz_gradient_accum, gaussians.denom, gaussians.max_radii2D]
    dev = tensors[0].device
    vals = torch.tensor([float(o["step"]) if k == "step" else float(o["lr"]) for k, o in scalars], dtype=torch.float64, device=dev)
    with torch.no_grad():
        flat = torch.cat([t.reshape(-1).to(torch.float32) for t in tensors])
        _check_same_layout(flat.numel(), dev, group)
        dist.broadcast(flat, src=src, group=group)
        dist.broadcast(vals, src=src, group=group)
        off = 0
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
    for v, (k, o) in zip(vals.tolist(), scalars):
        if k == "step":
            o["step"] = torch.tensor(float(v), dtype=torch.float32) if isinstance(o["step"], torch.Tensor) else v
        else:
            o["lr"] = float(v)
    return flat.numel() * 4
