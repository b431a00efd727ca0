#!/bin/bash
# usage (on the GPU box): bash tools/pmc_baq.sh [tag] -- where the wavefronts of the BAQ kernels spend their cycles, per dispatch.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r3}
OUTF=$R/gpurun_out/${TAG}_pmc_baq.txt
: > $OUTF
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  OUT=$R/gpurun_out/pmcbaq_tmp; rm -rf $OUT; mkdir -p $OUT
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $OUT -o p --output-format csv -- python3 $R/bench.py --mode baq --steps 3 --cpu-seconds 0 > $OUT/log 2>&1 || { echo "set failed: $set"; tail -3 $OUT/log; continue; }
  python3 - <<PY >> $OUTF
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("bcfgpu::","")
        if "baq" not in k: continue
        acc[k][r["Counter_Name"].replace("SQ_","")]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
for k in sorted(acc):
    print(k+" (sum over %d dispatches): "%max(n[(k,c)] for (kk,c) in n if kk==k)+" ".join("%s=%.2fM"%(c,v/1e6) for c,v in sorted(acc[k].items())))
PY
done
rm -rf $R/gpurun_out/pmcbaq_tmp
cat $OUTF
