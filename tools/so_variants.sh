#!/bin/bash
# Experiments on a GPU box with libraries built beforehand (bcftools_amd/variants/<name>.so, e.g. one object compiled with other flags):
# each is loaded in place of libbcfgpu.so (BCFGPU_SO, bcftools_amd/lib.py) for one bench run under rocprofv3 --stats; the kernels whose
# names match are listed.  The product library is never touched.
# usage: bash tools/so_variants.sh "<pattern>|<pattern>" "<bench args>"   -> gpurun_out/sovar.txt
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
PAT=$1; BARGS=$2
: > $R/gpurun_out/sovar.txt
for so in $R/bcftools_amd/variants/*.so; do
  name=$(basename $so .so)
  export BCFGPU_SO=$so
  cd /tmp; rm -rf /tmp/sv
  rocprofv3 --kernel-trace --stats -d /tmp/sv -o s --output-format csv -- python3 $R/bench.py $BARGS > /tmp/sv.log 2>&1 || { echo "$name: run failed" >> $R/gpurun_out/sovar.txt; tail -3 /tmp/sv.log; continue; }
  k=$(python3 -c "
import csv
for r in csv.DictReader(open('/tmp/sv/s_kernel_stats.csv')):
    n=r['Name']
    if any(p in n for p in '$PAT'.split('|')): print(n.split('(')[0].replace('void ','').replace('bcfgpu::',''), 'avg %.3f ms;' % (float(r['AverageNs'])/1e6), end=' ')
")
  v=$(python3 -c "
import json
for l in open('/tmp/sv.log'):
    if l.startswith('{'):
        d=json.loads(l); print('value %.4g %s, %.3f ms/step' % (d['value'], d['unit'], d.get('ms_per_step', 0)))
" 2>/dev/null | tail -1)
  echo "$name: $v; $k" >> $R/gpurun_out/sovar.txt
done
unset BCFGPU_SO
cat $R/gpurun_out/sovar.txt
