#!/bin/bash
# Kernel stats of the headline bench with glfgen as two launches (BCFGPU_GLFGEN_SPLIT=1): bash tools/prof_split.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-split}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
export BCFGPU_GLFGEN_SPLIT=${BCFGPU_GLFGEN_SPLIT:-1}
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python3 $R/bench.py --cpu-seconds 0 --cpu-all-cores 0 --extras 0 --steps 6 > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
grep -E "glfgen|combine|mcall" $OUT/stats/stats_kernel_stats.csv | cut -c1-200
