#!/bin/bash
# usage (GPU box): bash tools/dbg/layout_probe.sh  -- --mode wgs and --mode indel with the product library (twice each)
cd $GRAFT_REPO_ROOT
for mode in wgs indel wgs indel; do
  extra=""; [ $mode = wgs ] && extra="--steps 3 --warmup 1"
  python3 bench.py --mode $mode --cpu-seconds 0 $extra 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$mode value %.4g %s' % (d['value'], d['unit']), {k: d[k] for k in ('ms_per_step','split_ms') if k in d})"
done
