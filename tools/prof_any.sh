#!/bin/bash
# rocprofv3 kernel-trace + stats of an arbitrary python script (run ON the GPU box via gpurun).
# Usage: tools/prof_any.sh <tag> <script.py> [args...]
TAG=$1; shift
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 "$@" > $OUT/stdout.log 2>&1 || { tail -20 $OUT/stdout.log; exit 1; }
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -25 $OUT/kernel_stats.csv
tail -5 $OUT/stdout.log
