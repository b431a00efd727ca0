#!/bin/bash
# Kernel trace of the indel stage on a GPU box: bash tools/prof_indel.sh [tag]  -> gpurun_out/indel_<tag>/
set -e
TAG=${1:-r3}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/indel_$TAG
rm -rf $OUT && mkdir -p $OUT
python3 $R/bench.py --mode indel --steps 4 --cpu-seconds 0 --indel-callers 0 > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python3 $R/bench.py --mode indel --steps 4 --cpu-seconds 0 --indel-callers 0 > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
cp $OUT/stats/stats_kernel_stats.csv $OUT/${TAG}_indel_kernel_stats.csv
cut -c1-160 $OUT/${TAG}_indel_kernel_stats.csv | head -30
tail -1 $OUT/bench.json | cut -c1-1500
