#!/bin/bash
# usage (GPU box): bash tools/dbg/mcall_variants.sh -- mcall.hip built with other options, timed under rocprofv3 on the headline tile and on the configs[4] shape
cd $GRAFT_REPO_ROOT
for cfg in "" "--groups 4 --haploid-frac 0.25" "--groups 4"; do
  echo "== [$cfg]"
  bash tools/file_variants.sh mcall.hip mcall_kernel "--extras 0 --cpu-seconds 0 --cpu-all-cores 0 --steps 10 $cfg" "base:" "w4:-DMCALL_WAVES_GRP=4 -DMCALL_WAVES_HAP=4" "td:-DMCALL_TRIPLE_DIRECT=1" "tdw4:-DMCALL_TRIPLE_DIRECT=1 -DMCALL_WAVES_GRP=4 -DMCALL_WAVES_HAP=4" 2>&1 | grep -v "^$"
done
