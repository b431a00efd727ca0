#!/bin/bash
# On a GPU box: baq.hip built with -D switches, each build loaded in place of the product library (BCFGPU_SO) for the BAQ tests.
# usage: bash tools/dbg/baq_test_variants.sh "<pytest -k expression>" "name:flags" ...
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
K=$1; shift
ALL="glfgen combine mcall indel gap_prep baq overlap pileup gvcf gather capmapq draw api tables"
OBJS=""; for o in $ALL; do if [ $o = baq ]; then OBJS="$OBJS /tmp/bv.o"; else OBJS="$OBJS $R/bcftools_amd/csrc/$o.o"; fi; done
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  cd $R/bcftools_amd/csrc
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w -mllvm -amdgpu-sched-strategy=max-ilp $flags -c baq.hip -o /tmp/bv.o || { echo "$name: build failed"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/bv.so $OBJS -ldl
  cd $R
  for rep in 1 2; do
    BCFGPU_SO=/tmp/bv.so python3 -m pytest tests/test_gpu_baq.py -m gpu -q -k "$K" 2>&1 | tail -1 | sed "s/^/$name run $rep: /"
  done
done
