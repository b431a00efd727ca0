#!/bin/bash
# Kernel stats of the BAQ mode.  bash tools/prof_baq.sh [tag] -> gpurun_out/baq_<tag>/<tag>_baq_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r3}; shift
OUT=$R/gpurun_out/baq_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/p -o s --output-format csv -- python3 $R/bench.py --mode baq --steps 6 --cpu-seconds 0 "$@" > $OUT/bench.log 2>&1 || { echo failed; tail -3 $OUT/bench.log; exit 1; }
grep -v rocclr $OUT/p/s_kernel_stats.csv | head -12 | cut -c1-170 > $OUT/${TAG}_baq_kernel_stats.csv
cat $OUT/${TAG}_baq_kernel_stats.csv
grep '^{' $OUT/bench.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d.get('pool_form')))"
rm -rf $OUT/p
