#!/bin/bash
# Kernel timeline of the bench step (durations and the gaps between consecutive kernels of the one-stream loop), from a rocprofv3 kernel trace.
# Run ON the GPU box via gpurun.  Usage: tools/prof_gaps.sh <tag> [extra bench flags]
set -e
TAG=${1:-r04}; shift || true
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/gaps_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o bench -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --sustain-seconds 0 --settle-steps 300 --headline-only "$@" > $OUT/bench_stdout.log 2>&1 || { tail -20 $OUT/bench_stdout.log; exit 1; }
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 tools/trace_gaps.py $F 45 > $OUT/timeline.txt
find $OUT -type f ! -name timeline.txt ! -name bench_stdout.log -delete; find $OUT -type d -empty -delete
cat $OUT/timeline.txt
