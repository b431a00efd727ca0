#!/bin/bash
# usage (on the GPU box): bash tools/pmc_glf.sh [tag] -- the counters behind the round-3 glfgen experiments, per kernel name, for the
# fused kernel and for the two-launch variant (BCFGPU_GLFGEN_SPLIT=1): instruction counts, where the wavefronts wait, LDS conflicts.
# 4096-site tile, one timed step (counters of the warm-up and the timed dispatches summed: 3 dispatches of each kernel).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r3}
OUTF=$R/gpurun_out/${TAG}_pmc_glfgen_split.txt
: > $OUTF
for split in 0 1; do
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"; do
    OUT=$R/gpurun_out/pmcglf_tmp; rm -rf $OUT; mkdir -p $OUT
    BCFGPU_GLFGEN_SPLIT=$split timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $OUT -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 2 --sites 4096 --cpu-seconds 0 --cpu-all-cores 0 --extras 0 > $OUT/log 2>&1 || { echo "set failed: $set"; tail -3 $OUT/log; continue; }
    python3 - <<PY >> $OUTF
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("bcfgpu::","")
        if "glfgen" not in k: continue
        acc[k][r["Counter_Name"].replace("SQ_","")]+=float(r["Counter_Value"])
for k in sorted(acc):
    print("split=$split "+k+": "+" ".join("%s=%.2fM"%(c,v/1e6) for c,v in sorted(acc[k].items())))
PY
  done
done
cat $OUTF
