#!/bin/bash
# PMC passes over bench.py's describe launch (tools/time_clatch_batch.py) for the CLATCH kernel the environment selects
# (CLC_CLATCH_COPIES=4|8).  Run ON the GPU box: CLC_CLATCH_COPIES=8 tools/pmc_clatch_py.sh c8
TAG=${1:-c4}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/pmc_clatch_$TAG; rm -rf $OUT; mkdir -p $OUT
run() { local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 tools/time_clatch_batch.py > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; }
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$name" <<'PY'
import csv, sys, collections
f, name = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(float); cnt = collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    if "clatch" not in r["Kernel_Name"]: continue
    agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
print(name, {c: round(agg[c] / cnt[c], 1) for c in agg}, "launches", max(cnt.values()) if cnt else 0)
PY
  find $OUT/$name -type f -size +2M -delete
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run e SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD
run f GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_WAVES
