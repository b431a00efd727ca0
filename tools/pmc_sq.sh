#!/bin/bash
# usage (on the GPU box): bash tools/pmc_sq.sh  -- where the wavefronts of glfgen_kernel spend their cycles (product build),
# 4096-site tile, first dispatch: issue / wait split, instruction-fetch and LDS / vector-memory levels (level / instructions = latency).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_BUSY_CYCLES SQ_INSTS_BRANCH"; do
  tag=$(echo $set | cut -d' ' -f1)
  OUT=$R/gpurun_out/pmcsq_$tag; rm -rf $OUT; mkdir -p $OUT
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $OUT -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --sites 4096 --cpu-seconds 0 --cpu-all-cores 0 --extras 0 > $OUT/log 2>&1 || { echo "set failed: $set"; tail -3 $OUT/log; continue; }
  python3 - <<PY
import csv,glob
acc={}
first=None
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "glfgen" not in r["Kernel_Name"]: continue
        if first is None: first=r["Dispatch_Id"]
        if r["Dispatch_Id"]!=first: continue
        acc[r["Counter_Name"]]=acc.get(r["Counter_Name"],0)+float(r["Counter_Value"])
print(" ".join("%s=%.2fM"%(k.replace("SQ_",""),v/1e6) for k,v in sorted(acc.items())))
PY
done
