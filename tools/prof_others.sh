#!/bin/bash
# Kernel stats (rocprofv3 --kernel-trace --stats) of the secondary modes: BAQ / overlaps, the host-fed pileup, call -G with a ploidy array,
# the end-to-end region of --mode wgs (configs3_mixed).
# bash tools/prof_others.sh [tag] -> gpurun_out/others_<tag>/<tag>_{baq,pileup,mcall_grp}_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r3}
OUT=$R/gpurun_out/others_$TAG
rm -rf $OUT && mkdir -p $OUT
run() { name=$1; shift; rocprofv3 --kernel-trace --stats -d $OUT/$name -o s --output-format csv -- python3 $R/bench.py "$@" --cpu-seconds 0 --cpu-all-cores 0 --extras 0 > $OUT/$name.log 2>&1 || { echo "$name failed"; tail -3 $OUT/$name.log; return; }
  grep -E "^\"Name|bcfgpu::" $OUT/$name/s_kernel_stats.csv | head -16 | cut -c1-160 > $OUT/${TAG}_${name}_kernel_stats.csv; echo "== $name"; cat $OUT/${TAG}_${name}_kernel_stats.csv; }
run baq --mode baq --steps 3
run pileup --mode pileup --steps 4
run mcall_grp --groups 4 --haploid-frac 0.25 --steps 4
run wgs --mode wgs --baq 1 --steps 3 --warmup 1
