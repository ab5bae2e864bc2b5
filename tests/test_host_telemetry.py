"""bench.py's telemetry leg on the host (no GPU): tools/gpu_sampler.py's reduction of amdsmi samples to the fields the bench
line carries (`gpu_busy_in_timed_regions`, clock, power, PPT residency), and the start / stop plumbing of the sampler process."""
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _rows(t0, n, hz, busy, clk, power, second_gpu_busy=0):
    rows = []
    for k in range(n):
        rows.append({"t": t0 + k / hz, "gpus": [
            {"average_gfx_activity": second_gpu_busy, "current_gfxclks": [500] * 8, "current_socket_power": 200,
             "accumulation_counter": 1000 + k, "ppt_residency_acc": 0},
            {"average_gfx_activity": busy, "current_gfxclks": [clk - 10, clk + 10] * 4 + [65535], "current_socket_power": power,
             "accumulation_counter": 5000 + 10 * k, "ppt_residency_acc": 100 + 4 * k}]})
    return rows


def test_summarize_picks_the_busy_gpu_and_only_samples_inside_the_regions():
    from gpu_sampler import summarize
    t0 = 1000.0
    data = {"rows": _rows(t0, 40, 40.0, 5, 900, 300) + _rows(t0 + 1.0, 80, 40.0, 99, 2330, 1260) + _rows(t0 + 3.0, 40, 40.0, 3, 800, 250)}
    s = summarize(data, [[t0 + 1.0, t0 + 1.9], [t0 + 2.0, t0 + 2.9]])
    assert s["available"] and s["gpu_index"] == 1 and s["gpus_on_node"] == 2
    assert s["busy_pct_mean"] == 99.0 and s["gfx_activity_pct"]["min"] == 99
    assert s["sclk_mhz"]["median"] == 2330.0          # the 65535 "not populated" entry of the clock array is ignored
    assert s["socket_power_w"]["median"] == 1260
    assert abs(s["ppt_limit_residency_first_to_last_sample"] - 0.4) < 1e-9
    assert 70 <= s["samples"] <= 76


def test_summarize_says_unavailable_instead_of_guessing():
    from gpu_sampler import summarize
    assert summarize({"rows": [], "error": "amdsmi unavailable"}, [[0, 1]]) == {"available": False, "why": "amdsmi unavailable"}
    s = summarize({"rows": _rows(10.0, 5, 10.0, 50, 1000, 500)}, [[100.0, 101.0]])
    assert not s["available"] and s["samples_total"] == 5


def test_sampler_process_starts_before_the_gpu_and_stops_on_stdin_close():
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    s = bench.start_gpu_sampler(hz=50.0)
    assert s is not None
    time.sleep(0.3)
    out = bench.stop_gpu_sampler(s, [[time.time() - 1.0, time.time()]])
    assert s[0].returncode == 0 and not os.path.exists(s[1])
    assert "available" in out           # (this container has no amdsmi device: available False with the reason)
    if not out["available"]:
        assert out["why"]
