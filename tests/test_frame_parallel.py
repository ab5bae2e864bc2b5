"""N > 1 path on CPU: gloo, world_size 2 (the GPU path uses the same code with RCCL)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from splatloc_amd.frame_parallel import _shared_spans, allreduce_grads, shard_views, sync_densification_stats


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(100 + rank)
        P, C = 1000, 35
        grads = [torch.randn(P, 3, generator=g), None, torch.randn(P, C, generator=g), torch.randn(P, 1, generator=g),
                 torch.randn(P, 3, generator=g), torch.randn(P, 4, generator=g)[:, :3].t().contiguous().t()]
        assert not grads[-1].is_contiguous()
        local = [None if t is None else t.clone() for t in grads]
        allreduce_grads(grads, bucket_bytes=64 << 10)   # small buckets: exercises flat + single paths
        # reference: regenerate the other rank's tensors
        exp = []
        for r in range(world):
            gg = torch.Generator().manual_seed(100 + r)
            exp.append([torch.randn(P, 3, generator=gg), None, torch.randn(P, C, generator=gg),
                        torch.randn(P, 1, generator=gg), torch.randn(P, 3, generator=gg),
                        torch.randn(P, 4, generator=gg)[:, :3]])
        for i, t in enumerate(grads):
            if t is None:
                continue
            want = sum(e[i] for e in exp)
            assert torch.allclose(t, want, atol=1e-6), i
            assert not torch.equal(t, local[i])
        # gradients carved out of one allocation (what the rasterizer's backward returns) are
        # reduced in place as one span; the padding between pieces is left alone semantically
        flat = torch.full((3 * 10 + 2 + 10 + 2 + 10 * 4 + 7,), 100.0)
        a, b, c = flat[0:30].view(10, 3), flat[32:42].view(10, 1), flat[44:84].view(10, 4)
        outsider = flat[84:91]
        for t in (a, b, c):
            t.fill_(float(rank + 1))
        spans, rest = _shared_spans([a, b, c])
        assert len(spans) == 1 and not rest and spans[0].numel() == 84
        assert spans[0].data_ptr() == flat.data_ptr()
        allreduce_grads([a, None, b, c])
        assert torch.all(a == 3.0) and torch.all(b == 3.0) and torch.all(c == 3.0)
        assert torch.all(outsider == 100.0)
        # a far-apart pair in one allocation is NOT a span (what lies between is not ours)
        big = torch.zeros(100)
        u, v = big[0:10], big[50:60]
        spans, rest = _shared_spans([u, v])
        assert not spans and len(rest) == 2
        u.fill_(1.0), v.fill_(2.0)
        allreduce_grads([u, v])
        assert torch.all(u == 2.0) and torch.all(v == 4.0) and torch.all(big[10:50] == 0)
        acc = torch.full((P, 1), float(rank + 1))
        den = torch.ones(P, 1)
        rad = torch.arange(P, dtype=torch.float32) * (1 if rank == 0 else -1)
        sync_densification_stats(acc, den, rad)
        assert torch.all(acc == 3.0) and torch.all(den == 2.0)
        assert torch.equal(rad, torch.arange(P, dtype=torch.float32))
        out[rank] = 1
    finally:
        dist.destroy_process_group()


def test_allreduce_and_stats_gloo_world2():
    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert dict(out) == {0: 1, 1: 1}


def test_shard_views_is_a_partition():
    views = [7, 3, 9, 1, 4]                      # window_size = 5 (train_gaussians.py:183-195)
    for world in (1, 2, 4, 8):
        parts = [shard_views(views, r, world) for r in range(world)]
        assert sorted(v for p in parts for v in p) == sorted(views)
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_single_process_is_a_noop():
    t = torch.ones(4, 3)
    allreduce_grads([t, None])
    assert torch.all(t == 1)
