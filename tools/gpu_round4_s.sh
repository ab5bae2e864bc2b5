#!/bin/bash
mkdir -p gpurun_out/r4s
python tools/ab.py --no-parity base exp1 exp2 f2bfix base exp1 exp2 f2bfix > gpurun_out/r4s/ab_S2.txt 2>&1; cat gpurun_out/r4s/ab_S2.txt
