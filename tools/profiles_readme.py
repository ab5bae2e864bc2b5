#!/usr/bin/env python3
"""Regenerates profiles/README.md FROM the files under profiles/ — every number in it is read out of the file it is quoted
for (round 3's hand-written table drifted: the README said 4 492.6 us where the file said 4 192.16).  Descriptions without
numbers come from the table below; a file that is not listed there still gets a row ("(undescribed)").
usage: python tools/profiles_readme.py            (writes profiles/README.md)"""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")

WHAT = [   # (file name regex, description; {placeholders} are filled by the extractors below)
    (r"r\d+_kernel_stats\.txt$", "`rocprofv3 --kernel-trace --stats` of the default bench (S2; round 3 on: window-batched, one launch sequence per 5 views): {kstats}"),
    (r"r\d+_ref_layout_kernel_stats\.txt$", "the same for `bench.py --workload S2-ref-layout` (500k, 640x480, C = 4): {kstats}"),
    (r"r\d+_pmc_hbm\.json$", "separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of the same command; HBM bytes per launch = 2 x FETCH (gfx950 correction) + WRITE: {hbm}"),
    (r"r\d+.*pmc_sq_counters.*\.json$", "SQ counter passes (`tools/pmc_passes.sh`) of the compositing kernels: {sq}"),
    (r"r\d+_timeline\.txt$", "kernel timeline of one step of the default bench (start, duration, idle gap): {timeline}"),
    (r"r\d+_ref_layout_timeline\.txt$", "kernel timeline of one window at the reference layout: {timeline}"),
    (r"r\d+_refine_step_timeline\.txt$", "kernel timeline of one `color_refinement` iteration under rocprofv3 (the live figures are in `r04_refine_idle*.json`): {timeline}"),
    (r"r\d+_bench_under_rocprof\.json$", "the bench line printed by the profiled run itself: {bench}"),
    (r"r\d+_bench.*\.json$", "bench line: {bench}"),
    (r"r\d+_stage_.*\.json$", "`bench.py --stage ...` line: {bench}"),
    (r"r\d+_clocks\.json$", "shader clock / socket power / throttle residency sampled at ~20 Hz by a separate process (amdsmi library) while bench.py ran >= 10 s of S2 windows (`tools/clock_trace.py`): {clocks}"),
    (r"r\d+_clocks_trace\.json$", "the thinned sample trace behind the clock summary"),
    (r"r\d+_grad_bars.*\.json$", "distribution of the absolute tolerance the full-size window gradients NEED (tensor-scale and per-row), normal and accurate mode, oracle modes 0 / 1 (`tools/grad_bar_probe.py`): {gradbars}"),
    (r"r\d+_refine_idle.*\.json$", "live GPU idle of a `color_refinement` iteration (`tools/refine_idle.py`): {idle}"),
    (r"r\d+_hostprof_.*\.txt$", "cProfile of the host side of a refinement / map step (`tools/hostprof_steps.py`)"),
    (r"r\d+_ab_probes\.txt$", "A/B and timing-probe log of the round (one box per block)"),
    (r"r\d+_knn\.json$", "`distCUDA2` wall times, brute force vs exact grid"),
    (r"r\d+_scene_lists.*\.json$", "one `color_refinement` iteration on a RECONSTRUCTED room (list-length distribution, per-kernel table; `tools/scene_lists.py`; suffix = the forced variant): {scenelists}"),
    (r"r\d+_scene.*\.json$", "`bench.py --stage scene`: the whole reconstruction schedule as one run (suffix: `replica_scale` = 180 key-frames of a 600k-Gaussian room, `radix_front_end` = SPLATRASTER_FRONT_END=0): {bench}{scene}"),
    (r"r\d+_ab_.*\.json$", "`tools/ab.py`: interleaved same-box A/B of variant libraries, paired statistics: {ab}"),
    (r"r\d+_bwd_profile\.txt$", "per-phase cycle counters of the wide backward's waves (`tools/bwd_prof.py`, `-DSR_BWD_PROFILE` variant from `tools/patches/r05_variants.patch`)"),
    (r"r\d+_lone_wave\.json$", "one wave alone on its SIMD walking one list of N entries (`tools/lone_wave.py`): us per 1 000 list entries of the narrow forward, one-wave vs four-wave team"),
    (r"r\d+_perview_idle.*\.json$", "the literal per-view drop-in loop (5 x `GaussianRasterizer.__call__` + backward) against the window: wall vs GPU busy time (`tools/perview_idle.py`): {perview}"),
    (r"r\d+_map_idle.*\.json$", "live GPU idle and raster / non-raster kernel split of one `training.map_step` (`tools/map_idle.py`): {mapidle}"),
    (r"r\d+_rccl_contact\.json$", "`tools/rccl_contact.py` on one MI355X: a world-size-1 `nccl` (RCCL) process group drives every collective call of the frame-parallel path (in-place span SUM, MAX, reduce-scatter + all-gather, header, broadcast_model): {rccl}"),
    (r"r\d+_bench_force_process_group.*\.json$", "`bench.py --gpus 1 --force-process-group`: the BASELINE step with its collectives issued on a world-size-1 RCCL group: {bench}"),
    (r"traffic\.json$", "per-stage HBM bytes per launch that `bench.py` replays as `roofline.traffic` (recorded workload / launch mode inside)"),
    (r"valu\.json$", "VALU / MFMA / SALU wave-instructions, busy fractions of the two compositing kernels per launch (replayed by `bench.py` as `frame_valu` / `roofline_valu`)"),
    (r"r01_v1_first_.*", "round 1: the first correct pipeline (per-value DPP reductions, no reach masks)"),
]


def _load(path):
    try:
        with open(path) as f:
            txt = f.read().strip()
        try:
            return json.loads(txt)
        except ValueError:
            return json.loads(next(ln for ln in txt.splitlines() if ln.startswith("{")))
    except Exception:  # noqa: BLE001
        return None


def kstats(path):
    rows = []
    for ln in open(path):
        m = re.match(r"(composite_(?:bwd|fwd)_kernel<[^>]*>)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)", ln)
        if m:
            rows.append(f"`{m.group(1)}` {float(m.group(4)):.2f} us avg over {m.group(2)} launches ({float(m.group(5)):.1f} %)")
    return "; ".join(rows[:3]) or "(no compositing kernel rows)"


def hbm(path):
    j = _load(path) or {}
    out = []
    for k, e in j.items():
        if "composite" in k and isinstance(e, dict) and "hbm_bytes_per_launch" in e:
            out.append(f"`{k.split('<')[0]}<{k.split('<')[1][:14]}` {e['hbm_bytes_per_launch'] / 1e9:.2f} GB per launch")
    return "; ".join(out[:3]) or "(see file)"


def sq(path):
    j = _load(path) or {}
    out = []
    for k, c in j.items():
        if "composite" in k and "GRBM_GUI_ACTIVE" in c and "SQ_ACTIVE_INST_VALU" in c:
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0
            out.append(f"`{k[:34]}` VALU-busy {c['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * cyc):.2f}, "
                       f"{(c.get('SQ_INSTS_VALU', 0) - c.get('SQ_INSTS_MFMA', 0)) / 1e6:.0f} M VALU + {c.get('SQ_INSTS_MFMA', 0) / 1e6:.0f} M MFMA wave-instr per launch")
    return "; ".join(out[:2]) or "(see file)"


def timeline(path):
    last = [ln for ln in open(path) if "span" in ln]
    return last[-1].strip() if last else "(see file)"


def bench(path):
    j = _load(path)
    if not isinstance(j, dict) or "value" not in j:
        return "(see file)"
    s = f"{j['value']} {j.get('unit', '')}, {j.get('ms_per_step')} ms per step"
    if isinstance(j.get("repeats"), dict):
        s += f" (median of {j['repeats']['regions']} regions: {j['repeats']['frames_per_s_min']} – {j['repeats']['frames_per_s_max']})"
    r = j.get("roofline")
    if isinstance(r, dict) and r.get("avg_ms"):
        s += f"; dominant kernel `{r.get('kernel')}` {r['avg_ms']} ms per launch = {r.get('achieved')} GB/s = {r.get('frac')} of the HBM peak"
    wl = (j.get("config") or {}).get("workload", "")
    return s + (f" — {wl[:90]}" if wl else "")


def clocks(path):
    j = _load(path) or {}
    a = j.get("amdsmi_library_20hz") or {}
    c, p = a.get("gfxclk_mhz_mean_over_xcds") or {}, a.get("socket_power_w") or {}
    res = (a.get("residency_counters_first_last") or {})
    acc, ppt = res.get("accumulation_counter"), res.get("ppt_residency_acc")
    frac = f", power-limit (PPT) residency {100 * (ppt[1] - ppt[0]) / max(acc[1] - acc[0], 1):.0f} % of the samples' time" if acc and ppt else ""
    return (f"gfx clock mean over the 8 XCDs {c.get('median')} MHz median ({c.get('min')} – {c.get('max')}), socket power {p.get('median')} W median "
            f"of a {j.get('power_cap_w')} W cap{frac}, {a.get('samples')} samples at {a.get('sample_rate_hz')} Hz, bench {j.get('bench_value_frames_per_s')} frames/s")


def gradbars(path):
    j = _load(path) or {}
    out = []
    for c in j.get("cases", []):
        if c["rtol"] == 1e-4 and c["oracle_alpha_mode"] == 0:
            t = c["tensors"]["dL_dmeans3D"]
            out.append(f"{c['workload']} {'accurate' if c['deterministic'] else 'normal'} mode: dL/dmeans3D needs <= {t['by_tensor_max_q'][-1]:.1e} of the tensor max, "
                       f"{t['row_frac_above']['0.001']:.1e} of its elements > 1e-3 of their row max")
    return "; ".join(out) or "(see file)"


def idle(path):
    j = _load(path) or {}
    return (f"{j.get('workload')}: wall {j.get('wall_us_per_iteration')} us / iteration, GPU busy {j.get('gpu_busy_us_per_iteration_torch_profiler')} us, "
            f"idle {j.get('idle_us_per_iteration')} us, host enqueue {j.get('host_enqueue_us_per_iteration')} us")


def scenelists(path):
    j = _load(path) or {}
    top = ", ".join(f"{k['kernel'].split('(')[0].replace('void sr::', '').replace('sr::', '')[:28]} {k['us']}" for k in (j.get("kernels") or [])[:3]
                    if isinstance(k, dict) and "kernel" in k and "us" in k)
    return (f"{j.get('keyframes')} key-frames, {j.get('rows')} rows, front end {j.get('front_end')}: refinement {j.get('refine_us_per_iteration')} us / iteration"
            + (f"; largest kernels (us): {top}" if top else ""))


def scene(path):
    j = _load(path) or {}
    if "map_ms_per_iteration" not in j:
        return ""
    return (f"; map {j['map_ms_per_iteration']} ms / iteration, refinement {j.get('refine_ms_per_iteration')} ms / iteration, "
            f"{j.get('rows_final')} rows, peak {j.get('peak_memory_GB')} GB")


def ab(path):
    j = _load(path) or {}
    out = []
    for n, v in (j.get("variants") or {}).items():
        s_ = f"{n} {v['fps']['mean']:.1f} frames/s"
        d = (j.get("vs_base") or {}).get(n)
        if d:
            f = d["fps_diff"]
            s_ += f" ({f['mean']:+.2f}, CI {f['ci95'][0]:+.2f} .. {f['ci95'][1]:+.2f}{', significant' if f.get('significant_at_5pct') else ''})"
        if "sclk_mhz" in v:
            s_ += f" at {v['sclk_mhz']['mean']:.0f} MHz / {v['power_w']['mean']:.0f} W"
        out.append(s_)
    return f"{j.get('workload')}, {j.get('runs_per_variant')} runs per variant: " + "; ".join(out) if out else "(see file)"


def perview(path):
    j = _load(path) or {}
    keys = [k for k in j if isinstance(j[k], (int, float)) and ("us" in k or "ms" in k)][:6]
    return ", ".join(f"{k} {j[k]}" for k in keys) or "(see file)"


def mapidle(path):
    j = _load(path) or {}
    return (f"{j.get('workload')}, densify every {j.get('densify_every')}: wall {j.get('wall_us_per_step')} us / step, GPU busy "
            f"{j.get('gpu_busy_us_per_step_torch_profiler')} us (raster {j.get('raster_kernels_us_per_step')}, other {j.get('non_raster_kernels_us_per_step')}), "
            f"idle {j.get('idle_us_per_step')} us, {j.get('kernels_per_step')} kernels")


def rccl(path):
    j = _load(path) or {}
    return f"backend {j.get('backend')}, RCCL {j.get('rccl_version')}, HSA_ENABLE_IPC_MODE_LEGACY={((j.get('env') or {}).get('HSA_ENABLE_IPC_MODE_LEGACY'))}, ok = {j.get('ok')}"


EXTRACT = {"scenelists": scenelists, "scene": scene, "ab": ab, "perview": perview, "mapidle": mapidle, "rccl": rccl, "kstats": kstats, "hbm": hbm, "sq": sq, "timeline": timeline, "bench": bench, "clocks": clocks, "gradbars": gradbars, "idle": idle}


def describe(name, path):
    for pat, text in WHAT:
        if re.search(pat, name):
            for key, fn in EXTRACT.items():
                if "{" + key + "}" in text:
                    try:
                        text = text.replace("{" + key + "}", fn(path))
                    except Exception as ex:  # noqa: BLE001
                        text = text.replace("{" + key + "}", f"(unreadable: {ex!r})")
            return text
    return "(undescribed)"


def render() -> str:
    files = sorted((os.path.basename(p) for p in glob.glob(os.path.join(PROF, "*")) if not p.endswith("README.md")),
                   key=lambda n: (not n.startswith("r"), -(int(n[1:3]) if re.match(r"r\d\d_", n) else 0), n))
    lines = ["# profiles/ — rocprofv3 evidence, per round", "",
             "All numbers: MI355X (gfx950), `bench.py`.  **This table is generated** by `tools/profiles_readme.py` from the files it",
             "describes: every figure below is read out of the file in the same row.  Collected on the GPU box with",
             "`tools/gpu_profile.sh` / `tools/pmc_passes.sh` / `tools/clock_trace.py`, condensed by `tools/summarize_prof.py`, copied here by",
             "`tools/collect_profiles.py` (raw rocpd databases stay in the scratch directory `gpurun_out/`).", "",
             "| file | what |", "|---|---|"]
    for n in files:
        lines.append(f"| `{n}` | {describe(n, os.path.join(PROF, n))} |")
    return "\n".join(lines) + "\n"


def main():
    txt = render()
    with open(os.path.join(PROF, "README.md"), "w") as f:
        f.write(txt)
    print("wrote profiles/README.md:", txt.count("\n| `"), "files;", txt.count("(undescribed)"), "undescribed")


if __name__ == "__main__":
    main()
