#!/bin/bash
# usage (GPU box): bash tools/dbg/hwq_probe.sh  -- bench.py --mode wgs with the runtime's number of hardware queues at its default (4) and raised:
# do the realignment's stream chains share queues?
cd $GRAFT_REPO_ROOT
for q in "" 8 12; do
  [ -n "$q" ] && export GPU_MAX_HW_QUEUES=$q
  python3 bench.py --mode wgs --steps 3 --warmup 1 --cpu-seconds 0 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('GPU_MAX_HW_QUEUES=[$q] value %.4g ms/step %.2f split %s' % (d['value'], d['ms_per_step'], d['split_ms']))"
done
