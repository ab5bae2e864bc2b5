#!/usr/bin/env python3
"""Clock / power trace of the MI355X around a long run of the bench's S2 window ("measure the clock").

    python tools/clock_trace.py OUT.json [bench args]

Starts a SEPARATE monitoring process that samples the GPU's shader clock, power and (when exposed) throttle status at
>= 10 Hz — straight from sysfs (hwmon `freq1_input`, `power1_average` / `power1_input`, `pp_dpm_sclk`), with
`amd-smi metric --json` every `--smi-every` samples as a second, independent source — then runs
`bench.py --steps 300 --repeats 5 --no-cpu-baseline --no-multi-stream` (>= 10 s of back-to-back S2 windows) as a child
process, stops the monitor and writes the samples that fall inside the bench's timed regions (the bench line carries
their unix time stamps) with min / median / max.  This process never touches the GPU.  Nothing here needs root; every
source that is missing on the box is recorded as such instead of guessed.
"""
import glob
import json
import multiprocessing as mp
import os
import shutil
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def discover():
    """sysfs files of the first AMD GPU that has a hwmon node"""
    src = {}
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        if _read(os.path.join(card, "vendor")) not in ("0x1002",):
            continue
        hw = sorted(glob.glob(os.path.join(card, "hwmon", "hwmon*")))
        src = {"card": card, "hwmon": hw[0] if hw else None}
        for name in ("pp_dpm_sclk", "gpu_busy_percent", "current_link_speed"):
            p = os.path.join(card, name)
            if os.path.exists(p):
                src[name] = p
        if hw:
            for name in ("freq1_input", "freq2_input", "power1_average", "power1_input", "power1_cap", "temp1_input",
                         "temp2_input"):
                p = os.path.join(hw[0], name)
                if os.path.exists(p):
                    src[name] = p
        if hw:
            break
    return src


def _smi_sample():
    exe = shutil.which("amd-smi")
    if not exe:
        return None
    try:
        r = subprocess.run([exe, "metric", "-g", "0", "--clock", "--power", "--usage", "--json"], capture_output=True,
                           text=True, timeout=5)
        return json.loads(r.stdout) if r.returncode == 0 and r.stdout.strip() else {"rc": r.returncode, "err": r.stderr[-300:]}
    except Exception as ex:  # noqa: BLE001
        return {"error": repr(ex)}


def _lib_sampler():
    """The amdsmi Python binding (same library the amd-smi CLI uses): one `amdsmi_get_gpu_metrics_info` per GPU per sample
    is fast enough for >= 10 Hz.  Returns (sample_fn, n_gpus) or (None, 0)."""
    try:
        import amdsmi
        amdsmi.amdsmi_init()
        handles = amdsmi.amdsmi_get_processor_handles()
    except Exception:  # noqa: BLE001
        return None, 0
    keep = ("current_gfxclks", "current_gfxclk", "average_gfxclk_frequency", "current_socket_power", "average_socket_power",
            "average_gfx_activity", "average_umc_activity", "throttle_status", "indep_throttle_status", "temperature_hotspot",
            "accumulation_counter", "prochot_residency_acc", "ppt_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc",
            "hbm_thm_residency_acc", "gfx_below_host_limit_acc", "gfx_below_host_limit_ppt_acc", "gfx_below_host_limit_thm_acc",
            "gfx_below_host_limit_total_acc", "gfx_low_utilization_acc")

    def sample():
        rows = []
        for h in handles:
            try:
                m = amdsmi.amdsmi_get_gpu_metrics_info(h)
                rows.append({k: m[k] for k in keep if k in m})
            except Exception as ex:  # noqa: BLE001
                rows.append({"error": repr(ex)[:120]})
        return rows

    return sample, len(handles)


def monitor(stop, path, hz, smi_every):
    src = discover()
    rows, smi, lib_rows = [], [], []
    lib_sample, n_gpus = _lib_sampler()
    period = 1.0 / hz
    k = 0
    nxt = time.time()
    while not stop.is_set():
        row = {"t": time.time()}
        for name in ("freq1_input", "freq2_input", "power1_average", "power1_input", "gpu_busy_percent", "temp1_input"):
            if name in src:
                v = _read(src[name])
                try:
                    row[name] = int(v)
                except (TypeError, ValueError):
                    row[name] = v
        rows.append(row)
        if lib_sample is not None:
            lib_rows.append({"t": time.time(), "gpus": lib_sample()})
        if smi_every and k % smi_every == 0:
            smi.append({"t": time.time(), "metric": _smi_sample()})
        k += 1
        nxt += period
        d = nxt - time.time()
        if d > 0:
            time.sleep(d)
        else:
            nxt = time.time()
    with open(path, "w") as f:
        json.dump({"sources": src, "rows": rows, "amd_smi": smi, "amdsmi_lib": lib_rows, "amdsmi_lib_gpus": n_gpus}, f)


def _stats(vals):
    vals = [v for v in vals if isinstance(v, (int, float))]
    if not vals:
        return None
    return {"n": len(vals), "min": min(vals), "median": statistics.median(vals), "max": max(vals),
            "mean": round(statistics.fmean(vals), 2)}


def _smi_numbers(entry):
    """(sclk MHz list, power W, throttle text) out of one `amd-smi metric --json` record, whatever its exact shape"""
    out = {"gfx_clk_mhz": [], "power_w": None, "throttle": None}

    def walk(o, key=""):
        if isinstance(o, dict):
            if "clk" in o and isinstance(o["clk"], dict) and key.lower().startswith("gfx"):
                v = o["clk"].get("value")
                if isinstance(v, (int, float)):
                    out["gfx_clk_mhz"].append(v)
            for k, v in o.items():
                kl = k.lower()
                if kl in ("socket_power", "current_socket_power", "average_socket_power") and out["power_w"] is None:
                    vv = v.get("value") if isinstance(v, dict) else v
                    if isinstance(vv, (int, float)):
                        out["power_w"] = vv
                if "throttle" in kl and out["throttle"] is None and not isinstance(v, (dict, list)):
                    out["throttle"] = v
                walk(v, k)
        elif isinstance(o, list):
            for v in o:
                walk(v, key)

    walk(entry)
    return out


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "clocks.json")
    bench_args = sys.argv[2:] or ["--steps", "300", "--repeats", "5", "--warmup", "5", "--no-cpu-baseline", "--no-multi-stream"]
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    raw = out_path + ".raw"
    stop = mp.Event()
    mon = mp.Process(target=monitor, args=(stop, raw, 20.0, 20), daemon=True)
    mon.start()
    time.sleep(2.0)           # idle baseline
    t_start = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + bench_args, capture_output=True, text=True, cwd=ROOT)
    t_end = time.time()
    time.sleep(1.0)
    stop.set()
    mon.join(timeout=30)
    line = next((ln for ln in r.stdout.splitlines() if ln.startswith("{")), None)
    bench = json.loads(line) if line else None
    data = json.load(open(raw)) if os.path.exists(raw) else {"sources": {}, "rows": [], "amd_smi": []}
    spans = (bench or {}).get("timed_regions_unix") or [[t_start, t_end]]
    lo, hi = min(s[0] for s in spans), max(s[1] for s in spans)
    inside = [x for x in data["rows"] if any(a <= x["t"] <= b for a, b in spans)]
    idle = [x for x in data["rows"] if x["t"] < t_start]
    mhz = lambda rows, k: [x[k] / 1e6 for x in rows if isinstance(x.get(k), int)]   # noqa: E731  (hwmon freq in Hz)
    watts = lambda rows, k: [x[k] / 1e6 for x in rows if isinstance(x.get(k), int)]  # noqa: E731  (hwmon power in uW)
    # amdsmi library samples (20 Hz): the GPU this job runs on = the one with the highest mean activity inside the regions
    lib_in = [x for x in data.get("amdsmi_lib", []) if any(a <= x["t"] <= b for a, b in spans)]
    lib = None
    if lib_in:
        ng = len(lib_in[0]["gpus"])
        act = [statistics.fmean([x["gpus"][g].get("average_gfx_activity") or 0 for x in lib_in if g < len(x["gpus"])]) for g in range(ng)]
        g = max(range(ng), key=lambda i: act[i])
        mine = [x["gpus"][g] for x in lib_in]
        clk = []
        for m in mine:
            c = m.get("current_gfxclks")
            c = [v for v in c if isinstance(v, (int, float)) and 0 < v < 65535] if isinstance(c, list) else ([m["current_gfxclk"]] if isinstance(m.get("current_gfxclk"), (int, float)) else [])
            if c:
                clk.append({"min": min(c), "max": max(c), "mean": sum(c) / len(c)})
        acc = lambda k: ([mine[0].get(k), mine[-1].get(k)] if isinstance(mine[0].get(k), (int, float)) else None)  # noqa: E731
        lib = {"gpu_index_by_activity": g, "mean_activity_per_gpu": [round(a, 1) for a in act], "samples": len(mine),
               "sample_rate_hz": round(len(lib_in) / max(sum(b - a for a, b in spans), 1e-9), 1),
               "gfxclk_mhz_mean_over_xcds": _stats([c["mean"] for c in clk]), "gfxclk_mhz_min_over_xcds": _stats([c["min"] for c in clk]),
               "gfxclk_mhz_max_over_xcds": _stats([c["max"] for c in clk]),
               "socket_power_w": _stats([m.get("current_socket_power") if isinstance(m.get("current_socket_power"), (int, float)) and m.get("current_socket_power") < 65535 else m.get("average_socket_power") for m in mine]),
               "gfx_activity_pct": _stats([m.get("average_gfx_activity") for m in mine]),
               "temperature_hotspot_c": _stats([m.get("temperature_hotspot") for m in mine]),
               "throttle_status_values": sorted({str(m.get("throttle_status")) for m in mine}),
               "indep_throttle_status_values": sorted({str(m.get("indep_throttle_status")) for m in mine}),
               "residency_counters_first_last": {k: acc(k) for k in ("accumulation_counter", "prochot_residency_acc", "ppt_residency_acc",
                                                                       "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc",
                                                                       "gfx_below_host_limit_ppt_acc", "gfx_below_host_limit_thm_acc",
                                                                       "gfx_below_host_limit_total_acc", "gfx_low_utilization_acc")}}
    smi_in = [_smi_numbers(e["metric"]) for e in data["amd_smi"] if lo <= e["t"] <= hi and isinstance(e["metric"], (dict, list))]
    smi_clk = [max(s["gfx_clk_mhz"]) for s in smi_in if s["gfx_clk_mhz"]]
    summary = {
        "what": "shader clock / power sampled by a separate process (sysfs hwmon at 20 Hz + amd-smi metric once a second) while "
                "bench.py ran back-to-back S2 windows; statistics over the samples INSIDE the bench's timed regions",
        "bench_args": bench_args, "bench_rc": r.returncode,
        "bench_value_frames_per_s": (bench or {}).get("value"), "bench_ms_per_step": (bench or {}).get("ms_per_step"),
        "bench_repeats": (bench or {}).get("repeats"),
        "dominant_kernel_avg_ms": ((bench or {}).get("roofline") or {}).get("avg_ms"),
        "timed_seconds": round(sum(b - a for a, b in spans), 2), "samples_in_timed_regions": len(inside),
        "amdsmi_library_20hz": lib,
        "note_sysfs": "the hwmon node found first in sysfs (card0) is not necessarily the GPU this job was given: the box is an 8-GPU "
                      "node; the amdsmi figures are those of the GPU that was busy",
        "sources_found": data["sources"],
        "sclk_mhz_hwmon_freq1": _stats(mhz(inside, "freq1_input")), "sclk_mhz_idle_before": _stats(mhz(idle, "freq1_input")),
        "power_w_hwmon": _stats(watts(inside, "power1_average") or watts(inside, "power1_input")),
        "power_w_idle_before": _stats(watts(idle, "power1_average") or watts(idle, "power1_input")),
        "power_cap_w": (int(_read(data["sources"]["power1_cap"])) / 1e6) if data["sources"].get("power1_cap") and _read(data["sources"]["power1_cap"]) else None,
        "gpu_busy_percent": _stats([x.get("gpu_busy_percent") for x in inside]),
        "amd_smi_gfx_clk_mhz_max_over_xcds": _stats(smi_clk),
        "amd_smi_power_w": _stats([s["power_w"] for s in smi_in]),
        "amd_smi_throttle": sorted({str(s["throttle"]) for s in smi_in if s["throttle"] is not None}),
        "amd_smi_first_record_in_region": next((e["metric"] for e in data["amd_smi"] if lo <= e["t"] <= hi), None),
        "bench_stderr_tail": r.stderr[-400:] if r.returncode else None,
    }
    with open(out_path, "w") as f:
        json.dump(summary, f, indent=1)
    # a thinned copy of the trace itself (every 4th sample: 5 Hz) beside the summary
    thin = {"rows_5hz": data["rows"][::4], "amdsmi_lib_10hz": data.get("amdsmi_lib", [])[::2], "amd_smi": [{"t": e["t"], **_smi_numbers(e["metric"])} for e in data["amd_smi"]
                                                      if isinstance(e["metric"], (dict, list))]}
    with open(out_path.replace(".json", "_trace.json"), "w") as f:
        json.dump(thin, f)
    os.remove(raw)
    print(json.dumps({k: summary[k] for k in ("bench_value_frames_per_s", "amdsmi_library_20hz", "power_cap_w",
                                              "amd_smi_gfx_clk_mhz_max_over_xcds", "amd_smi_power_w", "amd_smi_throttle")}))


if __name__ == "__main__":
    main()
