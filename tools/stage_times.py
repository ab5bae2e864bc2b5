#!/usr/bin/env python3
"""Prints the per-stage avg ms from a bench.py JSON line (stdin)."""
import json
import sys
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    st = d["stages"]
    print(f"value={d['value']:.1f} fps ms/step={d['ms_per_step']:.3f} | " + " ".join(f"{k}={v['avg_ms']:.3f}" for k, v in st.items()))
