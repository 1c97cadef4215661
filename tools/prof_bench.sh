#!/bin/bash
# Round 5 profile set (run ON the GPU box via gpurun): rocprofv3 --kernel-trace --stats of the ONE-STREAM loop (where `roofline` is measured:
# every kernel has the device to itself) and of the default, pipelined command; then the separate --pmc passes over the one-stream loop.
set -e
TAG=${1:-r05}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
stats() { local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o bench -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --sustain-seconds 0 --settle-steps 500 --headline-only "$@" > $OUT/${name}_stdout.log 2>&1 || { tail -20 $OUT/${name}_stdout.log; exit 1; }
  find $OUT/$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${name}_kernel_stats.csv
  rm -rf $OUT/$name
  head -6 $OUT/${name}_kernel_stats.csv | cut -c1-200
}
stats one_stream --one-stream
stats pipelined
