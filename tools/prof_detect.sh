#!/bin/bash
# rocprofv3 over the detector alone (tools/time_detect_one.py: 600 calls of the two detector launches on one frame): kernel-trace + stats,
# then PMC passes (separate passes, --pmc only).  Run via gpurun.  Usage: tools/prof_detect.sh <tag> [W H]
TAG=${1:-r04}; shift
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/prof_detect_$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o p -- python3 tools/time_detect_one.py "$@" > $OUT/stdout.log 2>&1 || { tail -20 $OUT/stdout.log; exit 1; }
find $OUT/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -rf $OUT/kt
grep -E "Name|detect_" $OUT/kernel_stats.csv
tail -n 1 $OUT/stdout.log
run() { local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 tools/time_detect_one.py > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; }
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$name" <<'PY'
import csv, sys, collections
f, name = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:40]
    if "detect_" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in agg:
    print(name, k, {c: round(agg[k][c] / cnt[k][c], 1) for c in agg[k]}, "dispatches", max(cnt[k].values()))
PY
  rm -rf $OUT/$name
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq3 SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_IFETCH SQ_BUSY_CU_CYCLES
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
