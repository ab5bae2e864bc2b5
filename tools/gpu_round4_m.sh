#!/bin/bash
mkdir -p gpurun_out/r4m
O=gpurun_out/r4m
python tools/ab.py --no-parity base pf1 pf2 f2bfix base pf1 pf2 f2bfix > $O/ab_S2.txt 2>&1; cat $O/ab_S2.txt
python tools/ab.py --no-parity --workload S2-ref-layout base pf1 pf2 f2bfix > $O/ab_ref.txt 2>&1; cat $O/ab_ref.txt
