"""Frame-parallel data parallelism for SplatLoc's mapping step (one process per GPU).

In `SplatLoc.map` one optimisation step already sums the loss over `window_size` = 5
independent views before a single backward / optimizer step (train_gaussians.py:195-229,
265), so dealing the views of the window to ranks and SUM-all-reducing the parameter
gradients is mathematically the reference step.  Every rank holds a full
replica of the Gaussian scene; there is no collective on the data path of a single frame.

Collectives (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests):
  * SUM  of the parameter gradients (xyz, features, opacity, scaling, rotation, kp_score)
  * SUM  of the densification statistics increments (xyz_gradient_accum, denom;
         gaussian_model.py:677-679)
  * MAX  of max_radii2D (train_gaussians.py:240-244)
xGMI is point-to-point (7 links/GPU): the payload (64 B/Gaussian at the reference layout,
+4 B per extra feature channel) is sent as a few large buffers, never per-parameter-row.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_views(view_ids: Sequence[int], rank: int, world_size: int) -> List[int]:
    """Round-robin deal of the window's views to ranks (rank r gets views r, r+W, ...)."""
    return [v for i, v in enumerate(view_ids) if i % world_size == rank]


def allreduce_grads(tensors: Iterable[Optional[torch.Tensor]], group=None, bucket_bytes: int = 256 << 20):
    """SUM-all-reduce gradient tensors in place, coalesced into large flat buckets.

    Tensors that are None are skipped (a rank whose views saw no Gaussian still has dense
    zero grads from the rasterizer, so shapes agree across ranks).
    """
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    ts = [t for t in tensors if t is not None]
    bucket, size = [], 0
    pending = []

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
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


def sync_densification_stats(grad_accum_inc: torch.Tensor, denom_inc: torch.Tensor, max_radii2D: torch.Tensor,
                             group=None) -> None:
    """Make every replica densify identically: SUM the per-step increments of
    xyz_gradient_accum / denom and MAX max_radii2D (in place)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    packed = torch.cat([grad_accum_inc.reshape(-1), denom_inc.reshape(-1)])
    w1 = dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group, async_op=True)
    w2 = dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX, group=group, async_op=True)
    w1.wait()
    w2.wait()
    n = grad_accum_inc.numel()
    grad_accum_inc.copy_(packed[:n].view_as(grad_accum_inc))
    denom_inc.copy_(packed[n:].view_as(denom_inc))
