#!/bin/bash
# Kernel stats of the host-fed pileup mode.  bash tools/prof_pileup.sh [tag] [extra bench args] -> gpurun_out/pileup_<tag>/<tag>_pileup_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r3}; shift
OUT=$R/gpurun_out/pileup_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/p -o s --output-format csv -- python3 $R/bench.py --mode pileup --steps 4 "$@" > $OUT/bench.log 2>&1 || { echo failed; tail -3 $OUT/bench.log; exit 1; }
grep -v rocclr $OUT/p/s_kernel_stats.csv | head -14 | cut -c1-170 > $OUT/${TAG}_pileup_kernel_stats.csv
cat $OUT/${TAG}_pileup_kernel_stats.csv
tail -c 700 $OUT/bench.log
rm -rf $OUT/p
