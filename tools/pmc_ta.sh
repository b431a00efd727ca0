#!/bin/bash
# usage (on the GPU box): bash tools/pmc_ta.sh  -- texture-path counters of glfgen_kernel (product build), 4096-site tile, first dispatch
# (a pass with TA_FLAT_READ_WAVEFRONTS_sum / TA_*_STALLED_BY_TC_CYCLES_sum did not return on this pool: not in the list)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --list-avail > $R/gpurun_out/avail.txt 2>&1 || true
grep -o -E "\b(TA|TCP|TD|TCC|SQ|SQC)_[A-Za-z0-9_]+" $R/gpurun_out/avail.txt | sort -u > $R/gpurun_out/avail_names.txt
for set in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  OUT=$R/gpurun_out/pmcta_$tag; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc $set -d $OUT -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --sites 4096 --cpu-seconds 0 --cpu-all-cores 0 --extras 0 > $OUT/log 2>&1 || { echo "set failed: $set"; tail -3 $OUT/log; continue; }
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
print(" ".join("%s=%.2fM"%(k,v/1e6) for k,v in sorted(acc.items())))
PY
done
