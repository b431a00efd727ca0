#!/bin/bash
# usage (on the GPU box): bash tools/pmc_step.sh [tag] [bench args] -- counters of every kernel of the SNP step (glfgen, combine, mcall, i16,
# compaction) per dispatch: instruction counts, where the wavefronts wait, LDS and vector-memory activity.  8192-site tile by default.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r3}; shift
OUTF=$R/gpurun_out/${TAG}_pmc_step.txt
: > $OUTF
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA"; do
  OUT=$R/gpurun_out/pmcstep_tmp; rm -rf $OUT; mkdir -p $OUT
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $OUT -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --sites 8192 --cpu-seconds 0 --cpu-all-cores 0 --extras 0 "$@" > $OUT/log 2>&1 || { echo "set failed: $set"; tail -3 $OUT/log; continue; }
  python3 - <<PY >> $OUTF
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("bcfgpu::","")
        if not any(t in k for t in ("glfgen","combine","mcall","i16","compact","grp")): continue
        acc[k][r["Counter_Name"].replace("SQ_","")]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
for k in sorted(acc):
    print(k+" (per dispatch): "+" ".join("%s=%.2fM"%(c,v/1e6/n[(k,'SQ_'+c)]) for c,v in sorted(acc[k].items())))
PY
done
rm -rf $R/gpurun_out/pmcstep_tmp
cat $OUTF
