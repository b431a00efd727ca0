#!/bin/bash
# usage (GPU box): bash tools/dbg/hwq_probe2.sh  -- the other modes of bench.py with 4 and 8 hardware queues (twice each)
cd $GRAFT_REPO_ROOT
for mode in indel baq; do
  for q in 4 8 4 8; do
    GPU_MAX_HW_QUEUES=$q python3 bench.py --mode $mode --cpu-seconds 0 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$mode GPU_MAX_HW_QUEUES=$q value %.4g %s' % (d['value'], d['unit']), {k: d[k] for k in ('ms_per_step','split_ms','kernel_ms') if k in d})"
  done
done
