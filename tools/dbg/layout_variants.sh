#!/bin/bash
# usage (GPU box): bash tools/dbg/layout_variants.sh  -- indel.hip built with the launch layouts of launch_probaln_exact(), --mode wgs and --mode indel each
R=$GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
ALL="glfgen combine mcall indel gap_prep baq overlap pileup gvcf gather capmapq draw api tables"
PROD=$(make -s -C $R/bcftools_amd/csrc print-flags-indel)
OBJS=""; for o in $ALL; do if [ "$o" = indel ]; then OBJS="$OBJS /tmp/lv.o"; else OBJS="$OBJS $R/bcftools_amd/csrc/$o.o"; fi; done
for l in 0 2 3; do
  cd $R/bcftools_amd/csrc
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value $PROD -DPROBALN_LAYOUT=$l -c indel.hip -o /tmp/lv.o 2>/dev/null || { echo "layout $l: build failed"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lv.so $OBJS -ldl
  cd $R
  for mode in wgs indel; do
    extra=""; [ $mode = wgs ] && extra="--steps 3 --warmup 1"
    BCFGPU_SO=/tmp/lv.so python3 bench.py --mode $mode --cpu-seconds 0 $extra 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('layout $l $mode value %.4g %s' % (d['value'], d['unit']), d.get('split_ms', ''))"
  done
done
