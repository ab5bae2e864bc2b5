#!/usr/bin/env python3
"""Within-one-box A/B of variant libraries (tools/ablate.py): for every NAME given, runs the quick parity
subset and bench.py under SPLATRASTER_LIB=.../libsplatraster_NAME.so ("base" = the in-tree library) and prints
one line per variant.  usage: tools/ab.py [--no-parity] [--workload S2] base NAME1 NAME2 ..."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
parity = "--no-parity" not in args
workload = "S2"
if "--workload" in args:
    workload = args[args.index("--workload") + 1]
    del args[args.index("--workload"):args.index("--workload") + 2]
names = [a for a in args if not a.startswith("--")]
for name in names:
    env = dict(os.environ)
    if name != "base":
        env["SPLATRASTER_LIB"] = os.path.join(ROOT, "splatloc_amd", "_lib", "variants", f"libsplatraster_{name}.so")
    ok = "-"
    if parity:
        r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-x", "-q", "-k",
                            "forward_backward_parity or full_size_properties"], cwd=ROOT, env=env, capture_output=True, text=True)
        ok = "PASS" if r.returncode == 0 else "FAIL"
        if r.returncode:
            print(r.stdout[-1500:])
    r = subprocess.run([sys.executable, "bench.py", "--steps", "15", "--warmup", "4", "--no-cpu-baseline", "--workload", workload],
                       cwd=ROOT, env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(name, "bench failed", r.stderr[-800:])
        continue
    d = json.loads(line[0])
    st = d["stages"]
    print(f"{name:14s} parity={ok} fps={d['value']:.1f} | " + " ".join(f"{k}={v['avg_ms']:.4f}" for k, v in st.items()), flush=True)
