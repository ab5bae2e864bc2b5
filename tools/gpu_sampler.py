#!/usr/bin/env python3
"""GPU telemetry sampler for bench.py: a SEPARATE process, started by bench.py before anything in it touches
the GPU, that reads the amdsmi library's metrics table (activity %, per-XCD shader clocks, socket power, PPT residency
counters) at `hz` samples per second until its stdin closes, then writes the samples as JSON to `out` and exits.

    python tools/gpu_sampler.py OUT.json [hz]

It never loads HIP (amdsmi reads the driver's metrics table).  When the amdsmi binding is missing the file says so and holds
no samples: the bench reports the field as unavailable instead of guessing.  `summarize(samples, spans)` (imported by bench.py
and tools/ab.py) reduces the samples that fall inside the timed regions."""
import json
import os
import select
import statistics
import sys
import time

KEEP = ("average_gfx_activity", "average_umc_activity", "current_gfxclks", "current_gfxclk", "current_socket_power",
        "average_socket_power", "temperature_hotspot", "accumulation_counter", "ppt_residency_acc", "prochot_residency_acc",
        "socket_thm_residency_acc")


def _num(v):
    return v if isinstance(v, (int, float)) and 0 <= v < 65535 else None


def summarize(data: dict, spans) -> dict:
    """Statistics of the samples inside `spans` ([[t0, t1], ...] unix seconds).  On a multi-GPU node the GPU this job ran on is
    the one with the highest mean activity inside the spans."""
    if not data or not data.get("rows"):
        return {"available": False, "why": (data or {}).get("error", "no samples")}
    rows = [r for r in data["rows"] if any(a <= r["t"] <= b for a, b in spans)]
    if not rows:
        return {"available": False, "why": "no sample fell inside the timed regions", "samples_total": len(data["rows"])}
    ng = min(len(r["gpus"]) for r in rows)
    act = [statistics.fmean([_num(r["gpus"][g].get("average_gfx_activity")) or 0 for r in rows]) for g in range(ng)]
    g = max(range(ng), key=lambda i: act[i])
    mine = [r["gpus"][g] for r in rows]

    def stat(vals):
        vals = [v for v in vals if isinstance(v, (int, float))]
        if not vals:
            return None
        return {"n": len(vals), "min": round(min(vals), 1), "median": round(statistics.median(vals), 1), "max": round(max(vals), 1),
                "mean": round(statistics.fmean(vals), 2)}

    clk = []
    for m in mine:
        c = m.get("current_gfxclks")
        c = [v for v in c if _num(v)] if isinstance(c, list) else ([m["current_gfxclk"]] if _num(m.get("current_gfxclk")) else [])
        clk.append(sum(c) / len(c) if c else None)
    power = [(_num(m.get("current_socket_power")) or _num(m.get("average_socket_power"))) for m in mine]
    # the residency counters are cumulative: take them from the samples just outside each span too (first / last inside is enough)
    acc0, acc1 = mine[0].get("accumulation_counter"), mine[-1].get("accumulation_counter")
    ppt0, ppt1 = mine[0].get("ppt_residency_acc"), mine[-1].get("ppt_residency_acc")
    ppt = None
    if all(isinstance(v, (int, float)) for v in (acc0, acc1, ppt0, ppt1)) and acc1 > acc0:
        ppt = round((ppt1 - ppt0) / (acc1 - acc0), 3)
    busy = stat([_num(m.get("average_gfx_activity")) for m in mine])
    return {"available": True, "source": "amdsmi_get_gpu_metrics_info from a separate process (tools/gpu_sampler.py)",
            "samples": len(mine), "sample_rate_hz": round(len(mine) / max(sum(b - a for a, b in spans), 1e-9), 1),
            "gpu_index": g, "gpus_on_node": ng, "gfx_activity_pct": busy, "busy_pct_mean": busy["mean"] if busy else None,
            "sclk_mhz": stat(clk), "socket_power_w": stat(power), "umc_activity_pct": stat([_num(m.get("average_umc_activity")) for m in mine]),
            "hotspot_c": stat([_num(m.get("temperature_hotspot")) for m in mine]),
            "ppt_limit_residency_first_to_last_sample": ppt}


def main():
    out = sys.argv[1]
    hz = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
    rows, err = [], None
    try:
        import amdsmi
        amdsmi.amdsmi_init()
        handles = amdsmi.amdsmi_get_processor_handles()
    except Exception as ex:  # noqa: BLE001
        handles, err = [], f"amdsmi unavailable: {ex!r}"[:200]
    period = 1.0 / hz
    nxt = time.time()
    while True:
        if handles:
            row = {"t": time.time(), "gpus": []}
            for h in handles:
                try:
                    m = amdsmi.amdsmi_get_gpu_metrics_info(h)
                    row["gpus"].append({k: m[k] for k in KEEP if k in m})
                except Exception as ex:  # noqa: BLE001
                    row["gpus"].append({"error": repr(ex)[:80]})
            rows.append(row)
        nxt += period
        d = max(nxt - time.time(), 0.0)
        if d == 0.0:
            nxt = time.time()
        # stdin closing (the bench is done, or died) ends the sampling
        r, _, _ = select.select([sys.stdin], [], [], d)
        if r and not os.read(sys.stdin.fileno(), 4096):
            break
        if len(rows) > 200000:       # (a forgotten sampler must not grow without bound)
            break
    tmp = out + ".tmp"
    with open(tmp, "w") as f:
        json.dump({"hz": hz, "error": err, "rows": rows}, f)
    os.replace(tmp, out)


if __name__ == "__main__":
    main()
