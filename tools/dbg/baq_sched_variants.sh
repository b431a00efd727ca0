#!/bin/bash
# On a GPU box: baq.hip built with several scheduling strategies (and -D switches), each loaded in place of the product
# library (BCFGPU_SO) for `bench.py --mode baq` under rocprofv3 --stats.  -> gpurun_out/baq_sched.txt
# usage: bash tools/dbg/baq_sched_variants.sh "name:flags" ...
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$R/gpurun_out/baq_sched.txt; : > $OUT
ALL="glfgen combine mcall indel gap_prep baq overlap pileup gvcf gather capmapq draw api tables"
OBJS=""; for o in $ALL; do if [ $o = baq ]; then OBJS="$OBJS /tmp/bv.o"; else OBJS="$OBJS $R/bcftools_amd/csrc/$o.o"; fi; done
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  cd $R/bcftools_amd/csrc
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w $flags -c baq.hip -o /tmp/bv.o || { echo "$name: build failed" >> $OUT; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/bv.so $OBJS -ldl
  cd /tmp; rm -rf /tmp/bvp
  BCFGPU_SO=/tmp/bv.so rocprofv3 --kernel-trace --stats -d /tmp/bvp -o s --output-format csv -- python3 $R/bench.py --mode baq --steps 3 --warmup 1 --cpu-seconds 0 > /tmp/bv.log 2>&1 || { echo "$name: run failed" >> $OUT; tail -3 /tmp/bv.log; continue; }
  python3 - >> $OUT <<PY
import csv
o=[]
for r in csv.DictReader(open('/tmp/bvp/s_kernel_stats.csv')):
    n=r['Name']
    if 'baq_kernel' in n: o.append(n.split('(')[0].replace('void bcfgpu::','')+' max %.3f ms avg %.3f'%(float(r['MaxNs'])/1e6, float(r['AverageNs'])/1e6))
print("$name ($flags): "+"; ".join(sorted(o)))
PY
done
cat $OUT
