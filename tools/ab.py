#!/usr/bin/env python3
"""Decision-grade A/B of variant libraries (tools/ablate.py) on ONE box: INTERLEAVED runs of the default bench.

    tools/ab.py [--runs 12] [--workload S2] [--parity] [--out FILE.json] [--bench-args "..."] base NAME1 NAME2 ...

"base" = the in-tree library; NAME = splatloc_amd/_lib/variants/libsplatraster_NAME.so.  Round r runs every variant once, in
an order rotated by r (base is not always the first to meet a cold or a warm device), each run being `python bench.py
--no-cpu-baseline --no-multi-stream --workload W` — the default 5-region bench, value = its median region.  Reported per
variant: mean, standard deviation and the 95 % confidence interval of the mean (Student t) of frames/s and of the two
compositing stages; per variant against base: the mean of the PAIRED differences (same round), its 95 % interval and the
two-sided p-value of the paired t-test — a gain is real when the interval excludes 0.  Round 4 compared 3 + 3 un-interleaved
runs on different boxes; box-to-box spread is +-2 %, run-to-run on one box +-0.3 %: only same-box, interleaved, paired
comparisons resolve a 1 % effect.  --parity runs the quick parity subset once per variant first.
Every bench run carries its own telemetry (bench.py starts tools/gpu_sampler.py: amdsmi samples inside the timed regions), so
each variant also gets mean shader clock, socket power, PPT-limit residency and busy %, and the per-run scatter of (clock,
power) against (fwd ms, bwd ms) is kept: a forward / backward coupling that is a POWER effect shows up as a clock / power
difference between the variants."""
import json
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def t_crit(df):
    """two-sided 95 % quantile of Student's t (scipy when present, else a short table)"""
    try:
        from scipy import stats
        return float(stats.t.ppf(0.975, df))
    except Exception:  # noqa: BLE001
        tab = {1: 12.71, 2: 4.303, 3: 3.182, 4: 2.776, 5: 2.571, 6: 2.447, 7: 2.365, 8: 2.306, 9: 2.262, 10: 2.228, 11: 2.201,
               12: 2.179, 15: 2.131, 20: 2.086, 30: 2.042}
        return tab.get(df, 2.0)


def p_value(tval, df):
    try:
        from scipy import stats
        return float(2.0 * stats.t.sf(abs(tval), df))
    except Exception:  # noqa: BLE001
        return float("nan")


def summary(xs):
    n = len(xs)
    m = sum(xs) / n
    sd = math.sqrt(sum((x - m) ** 2 for x in xs) / (n - 1)) if n > 1 else 0.0
    half = t_crit(n - 1) * sd / math.sqrt(n) if n > 1 else float("nan")
    return {"n": n, "mean": round(m, 4), "sd": round(sd, 4), "ci95": [round(m - half, 4), round(m + half, 4)]}


def paired(a, b):
    d = [y - x for x, y in zip(a, b)]
    s = summary(d)
    n = len(d)
    tval = s["mean"] / (s["sd"] / math.sqrt(n)) if n > 1 and s["sd"] > 0 else float("nan")
    s["t"] = round(tval, 3) if tval == tval else None
    s["p_two_sided"] = round(p_value(tval, n - 1), 5) if tval == tval else None
    s["significant_at_5pct"] = bool(n > 1 and (s["ci95"][0] > 0 or s["ci95"][1] < 0))
    return s


def main():
    args = sys.argv[1:]

    def opt(name, default):
        if name in args:
            i = args.index(name)
            v = args[i + 1]
            del args[i:i + 2]
            return v
        return default

    runs = int(opt("--runs", "12"))
    workload = opt("--workload", "S2")
    out_path = opt("--out", None)
    extra = opt("--bench-args", "").split()
    parity = "--parity" in args
    names = [a for a in args if not a.startswith("--")]
    if len(names) < 2 or names[0] != "base":
        raise SystemExit(__doc__)

    def env_of(name):
        env = dict(os.environ)
        env.pop("SPLATRASTER_LIB", None)
        if name != "base":
            env["SPLATRASTER_LIB"] = os.path.join(ROOT, "splatloc_amd", "_lib", "variants", f"libsplatraster_{name}.so")
            if not os.path.exists(env["SPLATRASTER_LIB"]):
                raise SystemExit(f"missing variant library {env['SPLATRASTER_LIB']} (tools/ablate.py builds it)")
        return env

    res = {n: {"fps": [], "fwd_ms": [], "bwd_ms": [], "ms_per_step": [], "sclk_mhz": [], "power_w": [], "ppt_residency": [], "busy_pct": []}
           for n in names}
    par = {}
    if parity:
        for n in names:
            r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-x", "-q", "-k",
                                "forward_backward_parity or full_size_properties"], cwd=ROOT, env=env_of(n), capture_output=True, text=True)
            par[n] = "PASS" if r.returncode == 0 else "FAIL"
            print(f"parity {n}: {par[n]}", flush=True)
    for r_i in range(runs):
        order = names[r_i % len(names):] + names[:r_i % len(names)]
        for n in order:
            r = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-multi-stream", "--workload", workload] + extra,
                               cwd=ROOT, env=env_of(n), capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if not line:
                raise SystemExit(f"{n}: bench failed\n{r.stderr[-1500:]}")
            d = json.loads(line[0])
            res[n]["fps"].append(d["value"])
            res[n]["ms_per_step"].append(d["ms_per_step"])
            res[n]["fwd_ms"].append(d["stages"]["composite_fwd"]["avg_ms"])
            res[n]["bwd_ms"].append(d["stages"]["composite_bwd"]["avg_ms"])
            tel = d.get("gpu_telemetry_in_timed_regions") or {}
            if tel.get("available"):
                res[n]["sclk_mhz"].append((tel.get("sclk_mhz") or {}).get("mean"))
                res[n]["power_w"].append((tel.get("socket_power_w") or {}).get("mean"))
                res[n]["ppt_residency"].append(tel.get("ppt_limit_residency_first_to_last_sample"))
                res[n]["busy_pct"].append(tel.get("busy_pct_mean"))
        print(f"round {r_i + 1}/{runs}: " + "  ".join(f"{n} {res[n]['fps'][-1]:.1f}" for n in names), flush=True)
    out = {"workload": workload, "runs_per_variant": runs, "interleaved": True, "bench": "default 5-region bench (value = median region)",
           "parity": par or None, "variants": {}, "vs_base": {}}
    tel_keys = ("sclk_mhz", "power_w", "ppt_residency", "busy_pct")
    for n in names:
        clean = {k: [x for x in v if isinstance(x, (int, float))] for k, v in res[n].items()}
        out["variants"][n] = {k: summary(v) for k, v in clean.items() if v and (k not in tel_keys or len(v) == runs)}
        out["variants"][n]["fps_all"] = [round(x, 2) for x in res[n]["fps"]]
        if all(len(clean[k]) == runs for k in ("sclk_mhz", "power_w")):
            # the per-run scatter: does a run's kernel time follow its clock / power?
            out["variants"][n]["per_run"] = [{"fwd_ms": a, "bwd_ms": b, "sclk_mhz": c, "power_w": w}
                                             for a, b, c, w in zip(res[n]["fwd_ms"], res[n]["bwd_ms"], clean["sclk_mhz"], clean["power_w"])]
    for n in names[1:]:
        out["vs_base"][n] = {"fps_diff": paired(res["base"]["fps"], res[n]["fps"]),
                             "fwd_ms_diff": paired(res["base"]["fwd_ms"], res[n]["fwd_ms"]),
                             "bwd_ms_diff": paired(res["base"]["bwd_ms"], res[n]["bwd_ms"])}
        for k in ("sclk_mhz", "power_w", "ppt_residency"):
            a = [x for x in res["base"][k] if isinstance(x, (int, float))]
            b = [x for x in res[n][k] if isinstance(x, (int, float))]
            if len(a) == runs and len(b) == runs:
                out["vs_base"][n][k + "_diff"] = paired(a, b)
    txt = json.dumps(out, indent=1)
    if out_path:
        with open(out_path, "w") as f:
            f.write(txt + "\n")
    print(txt)
    print("\nvariant          frames/s mean (95 % CI)            fwd ms    bwd ms   sclk MHz  power W    paired d(frames/s) vs base (95 % CI)      p")
    for n in names:
        v = out["variants"][n]
        row = f"{n:14s} {v['fps']['mean']:9.2f} ({v['fps']['ci95'][0]:.2f} .. {v['fps']['ci95'][1]:.2f})   {v['fwd_ms']['mean']:8.4f}  {v['bwd_ms']['mean']:8.4f}"
        row += f"  {v['sclk_mhz']['mean']:8.1f}  {v['power_w']['mean']:7.1f}" if "sclk_mhz" in v and "power_w" in v else "         -        -"
        if n != "base":
            d = out["vs_base"][n]["fps_diff"]
            row += f"    {d['mean']:+7.2f} ({d['ci95'][0]:+.2f} .. {d['ci95'][1]:+.2f})  p = {d['p_two_sided']}  {'SIGNIFICANT' if d['significant_at_5pct'] else 'not significant'}"
        print(row)


if __name__ == "__main__":
    main()
