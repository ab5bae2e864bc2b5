#!/usr/bin/env python3
"""Register / LDS / spill summary of the kernels of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/regs.py composite_bwd [filter-substring]"""
import re
import subprocess
import sys

sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__file__), ".."))
from splatloc_amd import build  # noqa: E402

unit = sys.argv[1] + ".hip"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = [build._hipcc(), *build._flags(unit), "-c", f"{build.CSRC}/{unit}", "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for name, r in rows.items():
    if flt in name:
        print(f"{name:70s} VGPR {r.get('VGPRs', -1):4d} AGPR {r.get('AGPRs', 0):3d} SGPR {r.get('TotalSGPRs', -1):4d} "
              f"spillV {r.get('VGPRs Spill', 0):3d} scratch {r.get('ScratchSize', 0):4d} occ {r.get('Occupancy', -1)} LDS {r.get('LDS Size', 0)}")
