#!/bin/bash
mkdir -p gpurun_out/r4n
python tools/grad_bar_probe.py > gpurun_out/r4n/grad_bars.txt 2>&1
cp gpurun_out/r4_grad_bars.json gpurun_out/r4n/
grep -v Warn gpurun_out/r4n/grad_bars.txt | grep -A7 "mode 0 rtol 0.0001" | cut -c1-330
