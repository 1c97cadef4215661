#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench run (run ON the GPU box via gpurun).
# Usage: tools/prof_stats.sh <tag>
set -e
TAG=${1:-r02}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --sustain-seconds 0 --settle-steps 500 --headline-only > $OUT/bench_stdout.log 2>&1 || { tail -20 $OUT/bench_stdout.log; exit 1; }
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
# the raw trace is large and gpurun copies back at most 64 MiB: keep the summary only
find $OUT -type f ! -name kernel_stats.csv ! -name bench_stdout.log -delete; find $OUT -type d -empty -delete
cat $OUT/kernel_stats.csv | head -20
tail -2 $OUT/bench_stdout.log
