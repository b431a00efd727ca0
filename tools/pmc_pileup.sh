#!/bin/bash
# usage (on the GPU box): bash tools/pmc_pileup.sh [tag] -- where the wavefronts of the pileup kernels spend their cycles
# (issue / wait split, instruction counts, vector-memory and LDS activity), summed over the dispatches of one bench.py --mode pileup run.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r3}
OUT=$R/gpurun_out/pmcpileup_$TAG; rm -rf $OUT; mkdir -p $OUT
: > $OUT/${TAG}_pmc_pileup.txt
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_BUSY_CYCLES" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  D=$OUT/run; rm -rf $D
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $D -o p --output-format csv -- python3 $R/bench.py --mode pileup --steps 2 --sites 4096 > $OUT/log 2>&1 || { echo "set failed: $set"; tail -3 $OUT/log; continue; }
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob("$D/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("bcfgpu::","")
        if not k.startswith("pileup"): continue
        acc[k][r["Counter_Name"].replace("SQ_","")]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
with open("$OUT/${TAG}_pmc_pileup.txt","a") as o:
    for k in sorted(acc):
        line=k+" (per dispatch): "+" ".join("%s=%.3fM"%(c,v/1e6/max(1,n[(k,'SQ_'+c)] or n[(k,c)])) for c,v in sorted(acc[k].items()))
        print(line); o.write(line+"\n")
PY
done
rm -rf $OUT/run
