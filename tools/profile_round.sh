#!/bin/bash
# Full profile of bench.py for profiles/: kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes (each PMC pass in
# its own run with --kernel-trace only, as gpurun requires) + the SQ counter passes.  Raw rocprofv3 databases
# stay in /tmp (gpurun_out/ is limited to 64 MiB); only the summaries are written under gpurun_out/prof_TAG.
# usage (on the GPU box): tools/profile_round.sh TAG [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
RAW=/tmp/prof_$TAG
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $RAW/stats -o bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-multi-stream "$@" > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $RAW/fetch -o bench -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-multi-stream "$@" > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $RAW/write -o bench -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-multi-stream "$@" > /dev/null 2> $OUT/write.err
cd $R
python3 tools/summarize_prof.py $RAW $OUT/summary > $OUT/summary.txt
python3 tools/timeline.py $RAW/stats $OUT/bench_under_rocprof.json > $OUT/timeline.txt 2>&1
rm -f $OUT/*.err
