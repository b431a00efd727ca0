#!/bin/bash
# Kernel trace of `bench.py --mode wgs` on a GPU box: bash tools/prof_wgs.sh [tag] [extra bench args...]  -> gpurun_out/wgs_<tag>/
set -e
TAG=${1:-r5}; shift || true
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/wgs_$TAG
rm -rf $OUT && mkdir -p $OUT
python3 $R/bench.py --mode wgs --steps 3 --warmup 1 --cpu-seconds 0 "$@" > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python3 $R/bench.py --mode wgs --steps 3 --warmup 1 --cpu-seconds 0 "$@" > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
cp $OUT/stats/stats_kernel_stats.csv $OUT/${TAG}_wgs_kernel_stats.csv
cut -c1-200 $OUT/${TAG}_wgs_kernel_stats.csv | head -45
python3 - <<PY
import json
b=json.loads([l for l in open("$OUT/bench.json") if l.startswith("{")][-1])
print({k:b[k] for k in ("value","ms_per_step","split_ms","us_per_column","realignment")})
print(b["config"])
PY
