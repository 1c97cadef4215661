#!/bin/bash
# PMC passes over the CLATCH micro-benchmark binary (run via gpurun).  Usage: tools/pmc_clatch.sh <binary> <tag>
BIN=${1:-tools/clmb_BASE}; TAG=${2:-clatch}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
run() { local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o p -- $BIN > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; }
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$name" <<'PY'
import csv, sys, collections
f, name = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(float); cnt = collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    if "clatch" not in r["Kernel_Name"]: continue
    agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
print(name, {c: round(agg[c] / cnt[c], 1) for c in agg})
PY
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run b SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
# (a pass with eight TA_*_sum / TCP_*_sum counters hung rocprofv3 on this pool for 7 minutes in round 1: they are derived sums
#  over every TA / TCP instance, each instance using one of the block's few hardware counters -- see profiles/README.md;
#  request at most two of them per pass, in a pass of their own)
run e SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_IFETCH SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU
run f GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_WAVES
