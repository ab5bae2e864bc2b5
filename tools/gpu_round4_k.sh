#!/bin/bash
mkdir -p gpurun_out/r4k
O=gpurun_out/r4k
python tools/ab.py --no-parity base f2b f2bfix base f2b f2bfix > $O/ab_S2.txt 2>&1; cat $O/ab_S2.txt
python tools/ab.py --no-parity --workload S2-ref-layout base f2b f2bfix base f2b f2bfix > $O/ab_ref.txt 2>&1; cat $O/ab_ref.txt
python tools/ab.py --no-parity --workload S1 base f2b f2bfix > $O/ab_S1.txt 2>&1; cat $O/ab_S1.txt
