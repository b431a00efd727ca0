#!/bin/bash
# Collect SQ/LDS PMC counters for the three pipeline kernels on a GPU box (run through gpurun from the repo root):
#   bash tools/pmc.sh [sites]
# One rocprofv3 run per counter group (kernel-trace + pmc only); summary of the first dispatch of each kernel goes to
# gpurun_out/pmc/summary.csv.
set -e
SITES=${1:-4096}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
rm -rf $OUT && mkdir -p $OUT
groups=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU"
)
i=0
for g in "${groups[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $g -d $OUT/p$i -o p$i --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --sites $SITES --cpu-seconds 0 > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; exit 1; }
done
python3 - <<PY
import csv,glob,collections
rows=collections.OrderedDict()
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    seen=set()
    for r in csv.DictReader(open(f)):
        k=(r["Kernel_Name"].split("(")[0], r["Counter_Name"])
        d=r["Dispatch_Id"]
        # first dispatch of each kernel only
        first=rows.setdefault(("first",k[0],f), d)
        if d!=first: continue
        rows[k]=rows.get(k,0.0)+float(r["Counter_Value"])
with open("$OUT/summary.csv","w") as o:
    o.write("kernel,counter,value\n")
    for k,v in rows.items():
        if k[0]=="first": continue
        o.write('"%s",%s,%.0f\n'%(k[0],k[1],v))
print(open("$OUT/summary.csv").read())
PY
