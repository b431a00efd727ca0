#!/bin/bash
# usage (on the GPU box): bash tools/pmc_abl.sh mask ...   -- instruction counts of glfgen_kernel with parts switched off
# (diagnostics build, see tools/ablate_kernel.sh); 4096-site tile, first dispatch.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
make -s -C $R/bcftools_amd/csrc clean >/dev/null; make -s -j8 -C $R/bcftools_amd/csrc DIAG=1 >/dev/null 2>&1 || { echo "diag build failed"; exit 1; }
for a in "$@"; do
  OUT=$R/gpurun_out/pmcabl_$a; rm -rf $OUT; mkdir -p $OUT
  export BCFGPU_ABLATE=$a
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU -d $OUT -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --sites 4096 --cpu-seconds 0 --cpu-all-cores 0 --extras 0 > $OUT/log 2>&1 || { tail -5 $OUT/log; }
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
print("mask $a:", " ".join("%s=%.1fM"%(k.replace("SQ_",""),v/1e6) for k,v in sorted(acc.items())))
PY
done
unset BCFGPU_ABLATE
make -s -C $R/bcftools_amd/csrc clean >/dev/null; make -s -j8 -C $R/bcftools_amd/csrc >/dev/null 2>&1
