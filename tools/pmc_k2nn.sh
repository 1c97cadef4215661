#!/bin/bash
# PMC passes over the K2NN prototype / bench sweep (run via gpurun).  Usage: tools/pmc_k2nn.sh <tag> <kernel-substring> -- <program> [args]
TAG=$1; KSUB=$2; shift 3
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
run() { local name=$1; shift; local ctrs="$*"
  rocprofv3 --pmc $ctrs --output-format csv -d $OUT/$name -o p -- "${PROG[@]}" > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; }
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && KSUB="$KSUB" python3 - "$f" "$name" <<'PY'
import csv, sys, collections, os
f, name = sys.argv[1], sys.argv[2]
ks = os.environ["KSUB"]
agg = collections.defaultdict(float); cnt = collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    if ks not in r["Kernel_Name"]: continue
    agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
print(name, {c: round(agg[c] / cnt[c], 1) for c in agg}, "dispatches", max(cnt.values()) if cnt else 0)
PY
}
PROG=("$@")
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU
run c SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_VALU_MFMA_COEXEC_CYCLES
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
