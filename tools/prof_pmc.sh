#!/bin/bash
# rocprofv3 PMC passes over the default bench run (separate passes, --pmc only; run via gpurun).
# Usage: tools/prof_pmc.sh <tag>
TAG=${1:-r02}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
run() { # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --sustain-seconds 0 --settle-steps 0 --headline-only --one-stream > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; }
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$name" <<'PY'
import csv, sys, collections
f, name = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in agg:
    if "clc::" not in k: continue
    print(name, k, {c: round(agg[k][c] / cnt[k][c], 1) for c in agg[k]}, "dispatches", max(cnt[k].values()))
PY
  rm -rf $OUT/$name            # raw per-dispatch CSVs are large (gpurun copies back at most 64 MiB): the printed averages are the record
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
