#!/bin/bash
# usage (on the GPU box): bash tools/pmc_indel.sh [tag] -- where the wavefronts of the realignment kernels spend their cycles:
# issue / wait split and instruction counts per kernel name, summed over the dispatches of one bench.py --mode indel run.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r3}
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"; do
  tag=$(echo $set | cut -d' ' -f1)
  OUT=$R/gpurun_out/pmcindel_$TAG; rm -rf $OUT; mkdir -p $OUT
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $OUT -o p --output-format csv -- python3 $R/bench.py --mode indel --steps 1 --cpu-seconds 0 --indel-callers 0 > $OUT/log 2>&1 || { echo "set failed: $set"; tail -3 $OUT/log; continue; }
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("bcfgpu::","")
        if not (k.startswith("probaln") or k.startswith("gap_")): continue
        acc[k][r["Counter_Name"].replace("SQ_","")]+=float(r["Counter_Value"])
with open("$OUT/${TAG}_pmc_indel.txt","w") as o:
    for k in sorted(acc):
        line=k+": "+" ".join("%s=%.3fM"%(c,v/1e6) for c,v in sorted(acc[k].items()))
        print(line); o.write(line+"\n")
PY
done
