#!/bin/bash
# On a GPU box: baq.hip built with -D switches, each build loaded in place of the product library (BCFGPU_SO) for `bench.py --mode wgs`:
# the pool's BAQ stage in ms.  usage: bash tools/dbg/baq_pool_variants.sh "name:flags" ...
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
ALL="glfgen combine mcall indel gap_prep baq overlap pileup gvcf gather capmapq draw api tables"
OBJS=""; for o in $ALL; do if [ $o = baq ]; then OBJS="$OBJS /tmp/bv.o"; else OBJS="$OBJS $R/bcftools_amd/csrc/$o.o"; fi; done
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  cd $R/bcftools_amd/csrc
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w -mllvm -amdgpu-sched-strategy=max-ilp $flags -c baq.hip -o /tmp/bv.o || { echo "$name: build failed"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/bv.so $OBJS -ldl
  cd $R
  for rep in 1 2; do
    BCFGPU_SO=/tmp/bv.so python3 bench.py --mode wgs --steps 1 --warmup 1 --cpu-seconds 0 --baq 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$name run $rep: pool_baq %.2f ms' % d['front_ms']['pool_baq'])"
  done
done
