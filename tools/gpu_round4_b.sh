#!/bin/bash
mkdir -p gpurun_out/r4b
O=gpurun_out/r4b
python -m pytest tests/test_gpu_training.py -x -q 2>&1 | tail -15 > $O/training.txt; cat $O/training.txt
python tools/clock_trace.py $O/clocks.json > $O/clock_summary.txt 2>&1; cat $O/clock_summary.txt
python tools/grad_bar_probe.py > $O/grad_bars.txt 2>&1; tail -120 $O/grad_bars.txt
cp gpurun_out/r4_grad_bars.json $O/ 2>/dev/null
python bench.py --no-cpu-baseline --workload S2-ref-layout > $O/bench_ref.json 2> $O/bench_ref.err; cut -c1-300 $O/bench_ref.json
python bench.py --stage refine_step --workload S2-ref-layout --steps 200 --warmup 20 > $O/refine.json 2> $O/refine.err; cut -c1-400 $O/refine.json
python bench.py --stage map_step --workload S2-ref-layout --steps 100 --warmup 10 > $O/map_ref.json 2> $O/map_ref.err; cut -c1-400 $O/map_ref.json
