#!/bin/bash
# Full profile of bench.py for profiles/: kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes.
# usage (on the GPU box): tools/profile_round.sh TAG   -> gpurun_out/prof_TAG/{stats,fetch,write}
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o bench -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o bench -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/write.err
cd $R
python3 tools/summarize_prof.py gpurun_out/prof_$TAG gpurun_out/prof_$TAG/summary > $OUT/summary.txt
