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


def _reduce_step_worker(rank, world, port, out):
    """frame_parallel.reduce_step: the whole exchange of a step in TWO collectives, in place when the gradients and the
    statistics tail are one allocation (rasterizer.window_grad_span), packed otherwise; a rank WITHOUT views contributes
    zeros in the same layout (world size > window size: ranks 5..7 of 8 on a 5-view window)."""
    from splatloc_amd.frame_parallel import reduce_step
    from splatloc_amd.rasterizer import window_grad_layout, window_grad_span
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P, C = 1001, 4                         # odd P: alignment padding between the pieces
        shapes, offs, total = window_grad_layout(P, C)
        assert list(shapes) == ["m3", "op", "col", "sca", "rot"] and total % 4 == 0
        calls = []
        real = dist.all_reduce

        def counting(*a, **k):
            calls.append(k.get("op", a[1] if len(a) > 1 else None))
            return real(*a, **k)

        dist.all_reduce = counting
        try:
            # rank 0 "rendered" (values 1..), rank 1 had no views: zeros in the same layout
            sp = window_grad_span(P, C, "cpu", tail=True, zero=(rank == 1))
            assert sp["flat"].numel() == total + 2 * P and sp["tail"].shape == (2, P, 1)
            assert sp["tail"].data_ptr() == sp["flat"].data_ptr() + 4 * total
            if rank == 0:
                for k in ("m3", "op", "col", "sca", "rot"):
                    sp[k].fill_(2.0)
                sp["tail"][0].fill_(0.5)
                sp["tail"][1].fill_(1.0)
            grads = [sp[k] for k in ("m3", "col", "op", "sca", "rot")]
            max_r = torch.full((P,), float(rank * 7))
            seen = torch.zeros(P)
            seen[rank::2] = 1.0
            reduce_step(grads, [sp["tail"][0], sp["tail"][1]], [max_r], None)
            n_first = len(calls)
            calls.clear()
            for k in ("m3", "op", "col", "sca", "rot"):
                sp[k].fill_(2.0 if rank == 0 else 0.0)
            sp["tail"][0].fill_(0.5 if rank == 0 else 0.0)
            sp["tail"][1].fill_(1.0 if rank == 0 else 0.0)
            g, e, info = reduce_step(grads, [sp["tail"][0], sp["tail"][1]], [max_r, seen], None)
            assert info["collectives"] == 2 and info["sum_path"] == "in-place span", info
            # SUM + MAX + the 32-byte length header, on EVERY call (a rank that skipped the header because it had seen the
            # length before would pair its SUM with a diverging peer's header: round-4 advisor finding)
            assert len(calls) == 3 and n_first == 3 and info["header_collectives"] == 1, (calls, n_first)
            assert all(a is b for a, b in zip(g, grads))
            assert all(torch.all(t == 2.0) for t in g) and torch.all(e[0] == 0.5) and torch.all(e[1] == 1.0)
            assert torch.all(max_r == 7.0) and torch.all(seen == 1.0)
            assert info["sum_bytes"] == 4 * (total + 2 * P)
            calls.clear()
            # gradients that are NOT one allocation (map_step: they went through the fused activations' backward) are
            # packed by one cat and handed back as views of the reduced buffer
            gs = [torch.full((P, 3), float(rank + 1)), torch.full((P, 1, 3), float(rank + 1)), torch.zeros(P, 0, 3),
                  torch.full((P, 1), float(rank + 1))]
            inc = torch.full((2, P, 1), float(rank))
            g2, e2, info2 = reduce_step(gs, [inc[0], inc[1]], [max_r], None)
            assert info2["collectives"] == 2 and info2["sum_path"] == "packed"
            assert all(torch.all(t == 3.0) for t in g2 if t.numel()) and g2[2].shape == (P, 0, 3)
            assert g2[1].shape == (P, 1, 3) and torch.all(e2[0] == 1.0)
            assert g2[0].untyped_storage().data_ptr() == g2[3].untyped_storage().data_ptr()
            # the same exchange as reduce-scatter + all-gather (gloo has no reduce-scatter: emulated by the all-reduce; the
            # padding / packing of a length that is not a multiple of the world size is what this covers)
            odd = [torch.full((7, 3), float(rank + 1)), torch.full((4,), float(rank + 1))]      # 25 elements, world 2
            g3, _, info3 = reduce_step(odd, [], [max_r], None, mode="rs_ag")
            assert info3["mode"] == "rs_ag" and info3["sum_path"] == "packed" and info3["sum_bytes"] == 4 * 26
            assert all(torch.all(t == 3.0) for t in g3) and g3[0].shape == (7, 3)
            # ranks that disagree on the layout are refused, not summed — also when one of them has reduced "its" length
            # before (rank 0 repeats the length of the very first call, rank 1 brings a new one)
            bad = list(grads) + [sp["tail"][0], sp["tail"][1]] if rank == 0 else [torch.zeros(11)]
            try:
                reduce_step(bad, [], [max_r], None)
                raised = False
            except RuntimeError as ex:
                raised = "disagree" in str(ex)
            assert raised
        finally:
            dist.all_reduce = real
        out[rank] = 1
    finally:
        dist.destroy_process_group()


def test_reduce_step_two_collectives_gloo_world2():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_reduce_step_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert dict(out) == {0: 1, 1: 1}


def test_reduce_step_single_process_is_a_noop():
    from splatloc_amd.frame_parallel import reduce_step
    t, e, m = torch.ones(4, 3), torch.ones(4, 1), torch.ones(4)
    g, x, info = reduce_step([t], [e], [m])
    assert g[0] is t and x[0] is e and info["collectives"] == 0


def test_empty_window_is_not_an_error():
    """A rank without views (world size > window size) renders nothing: render_window returns ([], []) before any launch
    and rasterize_window([]) returns [] (round-3 advisor finding: IndexError on settings[0], the other ranks hang in the
    collective)."""
    import types
    from splatloc_amd.fused import render_window
    from splatloc_amd.rasterizer import _window_compatible, rasterize_window
    assert _window_compatible([]) is True
    assert rasterize_window([], torch.zeros(3, 3), [], torch.zeros(3, 4), torch.zeros(3, 1), scales=torch.zeros(3, 3),
                            rotations=torch.zeros(3, 4)) == []
    pc = types.SimpleNamespace(_xyz=torch.zeros(3, 3))      # CPU tensors: nothing may be launched (no CPU fallback exists)
    assert render_window([], pc, types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False),
                         torch.zeros(3)) == ([], [])


def test_bench_gpus_flag_launches_ranks_or_refuses(monkeypatch):
    """bench.py --gpus N without a launcher starts N ranks as a child process (round-3 verdict: args.gpus was never read);
    with RCCL and fewer GPUs than ranks it refuses (exit code 2) instead of measuring fewer GPUs than it reports."""
    import importlib.util
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_run(cmd, env=None, cwd=None, stdout=None):
        seen["cmd"], seen["env"], seen["stdout"] = cmd, env, stdout
        return types.SimpleNamespace(returncode=0)

    import types
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setenv("SPLATLOC_DIST_BACKEND", "gloo")
    assert bench.launch_ranks(2, ["--gpus", "2", "--steps", "3"]) == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "2", "--steps", "3"]
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert seen["stdout"] is sys.stdout      # the ranks write their JSON line to the launcher's REAL stdout (main() points fd 1 at stderr)
    monkeypatch.setenv("SPLATLOC_DIST_BACKEND", "nccl")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    seen.clear()
    assert bench.launch_ranks(8, ["--gpus", "8"]) == 2 and not seen
    # under a launcher whose world size is not what --gpus says, the run is refused
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    monkeypatch.undo()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_group_of_one_exchanges_nothing_unless_forced():
    """A process group of ONE rank: every function returns early (no collective) — unless `force=True` (or
    SPLATLOC_FORCE_COLLECTIVES=1), which issues every collective anyway: the first-contact path tests/test_gpu_rccl.py drives on
    RCCL, here on gloo.  SUM / MAX / broadcast over one rank leave the values unchanged."""
    from splatloc_amd import frame_parallel as fp
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        assert not fp.collectives_active() and fp.collectives_active(force=True)
        flat = torch.arange(100, dtype=torch.float32)
        a, b, c = flat[0:30].view(10, 3), flat[32:42].view(10, 1), flat[44:84].view(10, 4)
        tail = flat[84:100].view(2, 8)
        mx = torch.rand(7)
        ref = flat.clone(), mx.clone()
        g, e, info = fp.reduce_step([a, b, c], sum_extras=[tail[0], tail[1]], max_extras=[mx])
        assert info["collectives"] == 0 and g[0] is a
        for mode in ("ring", "rs_ag"):
            g, e, info = fp.reduce_step([a, b, c], sum_extras=[tail[0], tail[1]], max_extras=[mx], mode=mode, force=True)
            assert info["sum_path"] == "in-place span" and info["collectives"] == 2 and info["header_collectives"] == 1
            assert info["header_ms"] >= 0.0
            assert g[0].data_ptr() == a.data_ptr() and torch.equal(flat, ref[0]) and torch.equal(mx, ref[1])
        # packed path (pieces of different allocations)
        x, y = torch.randn(13), torch.randn(5, 1)
        rx, ry = x.clone(), y.clone()
        g, e, info = fp.reduce_step([x], sum_extras=[y], mode="rs_ag", force=True)
        assert info["sum_path"] == "packed" and torch.equal(g[0], rx) and torch.equal(e[0], ry)
        assert fp.allreduce_grads([a, b, c]) is None
        path = fp.allreduce_grads([a, None, b, c, torch.randn(4, 3)[:, :2]], force=True)
        assert path["spans"] == 1 and path["buckets"] == 1 and torch.equal(flat, ref[0])
        fp.FORCE_COLLECTIVES = True
        try:
            assert fp.collectives_active()
            g, e, info = fp.reduce_step([a, b, c], sum_extras=[tail[0], tail[1]], max_extras=[mx])
            assert info["collectives"] == 2
        finally:
            fp.FORCE_COLLECTIVES = False
    finally:
        dist.destroy_process_group()
