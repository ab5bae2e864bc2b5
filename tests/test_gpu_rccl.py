"""First contact with RCCL on the one GPU of the test box (BASELINE config 5 replaces
/root/reference/replica.sh:1-6): a WORLD-SIZE-1 "nccl" process group created with `device_id=` in a FRESH child process
drives every collective call of the N > 1 path — the in-place span SUM on a real window backward's allocation, the MAX,
the aliased reduce-scatter + all-gather pair of `--reduce rs_ag`, the 32-byte header, `broadcast_model`, `barrier`,
`destroy_process_group` — with the one-rank early returns bypassed (`force=True`).  Values must come back unchanged.

What this does not cover (one GPU): peer access over xGMI, more than one communicator rank."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _child_env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "SPLATLOC_DIST_BACKEND", "SPLATLOC_FORCE_COLLECTIVES"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _one_json_line(r, only_line=False):
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    if only_line:   # bench.py: stdout carries the JSON line and NOTHING else (RCCL / gloo banners go to stderr)
        assert r.stdout.strip().splitlines() == lines, r.stdout[:600]
    return json.loads(lines[0])


def test_rccl_world_size_one_drives_every_collective_of_the_frame_parallel_path():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_contact.py")], cwd=ROOT, env=_child_env(),
                       capture_output=True, text=True, timeout=600)
    out = _one_json_line(r)
    print("rccl_contact:", json.dumps(out))
    assert out["ok"] and out["backend"] == "nccl" and out["world_size"] == 1
    assert out["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    ran = out["ran"]
    assert ran["window_span_ring"]["sum_path"] == "in-place span" and ran["window_span_ring"]["collectives"] == 2
    # rs_ag on RCCL: reduce-scatter + all-gather + MAX (gloo would say 2: it emulates the pair with an all-reduce)
    assert ran["window_span_rs_ag"]["sum_path"] == "in-place span" and ran["window_span_rs_ag"]["collectives"] == 3
    assert ran["packed_ring"] == 2 and ran["packed_rs_ag"] == 3
    assert [s["collectives"] for s in ran["map_steps"]] == [2, 3]
    assert ran["broadcast_model_bytes"] > 0 and ran["rows_after_densify"] > 0


@pytest.mark.parametrize("reduce", ["ring", "rs_ag"])
def test_bench_force_process_group_runs_the_step_through_rccl(reduce):
    """`bench.py --gpus 1 --force-process-group`: the BASELINE step with a group of one — `init_process_group("nccl",
    device_id=)`, the span SUM in place in the backward's allocation, the MAX, the barrier + MAX-over-ranks timing."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-process-group", "--steps", "3", "--warmup", "1",
           "--workload", "S0", "--no-cpu-baseline", "--no-multi-stream", "--repeats", "2", "--reduce", reduce]
    env = dict(_child_env(), NCCL_DEBUG="VERSION")      # RCCL then prints a five-line banner per process group: it must not reach stdout
    out = _one_json_line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600), only_line=True)
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1 and out["dist_backend"] == "nccl"
    assert out["reduce_path"] == "in-place span" and out["collectives_per_step"] == (3 if reduce == "rs_ag" else 2)
    assert out["dist_env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and out["dist_env"]["rccl_version"]
    assert out["value"] > 0 and out["config"]["frames_per_step"] == 5


@pytest.mark.parametrize("replicas", [True, False])
def test_bench_scene_through_a_group_of_one(replicas):
    """`bench.py --stage scene --gpus 1 [--replicas] --force-process-group`: the scene schedule with the process group alive —
    replicas: barrier + all_gather_object only; frame-parallel: two collectives per map step + the broadcast after the
    refinement, all on RCCL."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--stage", "scene", "--gpus", "1", "--force-process-group", "--keyframes", "4",
           "--refine", "30", "--truth", "20000"] + (["--replicas"] if replicas else [])
    out = _one_json_line(subprocess.run(cmd, cwd=ROOT, env=_child_env(), capture_output=True, text=True, timeout=900))
    cfg = out["config"]
    assert cfg["multi_gpu_mode"] == ("replicas" if replicas else "frame-parallel") and cfg["dist_backend"] == "nccl"
    assert cfg["dist_env"]["world_size"] == 1
    assert (cfg["collectives_on_the_data_path"] == 0) == replicas
    assert out["rows_final"] > 0 and out["eval"]["mean_psnr"] > 5
