#!/bin/bash
# PMC passes over tools/bin/clatch_lab (run via gpurun).  Usage: tools/pmc_lab.sh <variant filter> <n> <tag>
FILT=${1:-production}; N=${2:-20000}; TAG=${3:-lab}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
run() { local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o p -- ./tools/bin/clatch_lab "$FILT" $N 10 > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; }
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$name" <<'PY'
import csv, sys, collections
f, name = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in agg:
    print(name, k, {c: round(agg[k][c] / cnt[k][c], 1) for c in agg[k]}, "dispatches", max(cnt[k].values()))
PY
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run b SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU
run c SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQC_TC_INST_REQ SQC_TC_STALL SQC_DCACHE_REQ
run e SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES
run f GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_WAVES
