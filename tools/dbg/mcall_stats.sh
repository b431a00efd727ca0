#!/bin/bash
# usage (GPU box): bash tools/dbg/mcall_stats.sh "<bench args>" ...  -- per-kernel averages of the caller's kernels under rocprofv3 for each argument set
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "$@"; do
  rm -rf /tmp/ms
  rocprofv3 --kernel-trace --stats -d /tmp/ms -o s --output-format csv -- python3 $R/bench.py --extras 0 --cpu-seconds 0 --cpu-all-cores 0 --steps 10 $cfg > /tmp/ms.log 2>&1
  echo "== [$cfg]"
  python3 -c "
import csv
for r in csv.DictReader(open('/tmp/ms/s_kernel_stats.csv')):
    n=r['Name']
    if any(p in n for p in ('mcall','grp_','i16','combine','compact')): print('  %-70s calls %s avg %.3f ms' % (n.split('(')[0].replace('void ','').replace('bcfgpu::','')[:70], r['Calls'], float(r['AverageNs'])/1e6))
"
done
