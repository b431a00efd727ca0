#!/bin/bash
# per-launch grid sizes and durations of the BAQ kernels of one `bench.py --mode baq` run -> gpurun_out/baq_trace.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/bt
rocprofv3 --kernel-trace -d /tmp/bt -o s --output-format csv -- python3 $R/bench.py --mode baq --steps 2 --warmup 1 --cpu-seconds 0 --cpu-all-cores 0 --extras 0 > /tmp/bt.log 2>&1 || { tail -5 /tmp/bt.log; exit 1; }
python3 - <<'PY' > $R/gpurun_out/baq_trace.txt
import csv, glob
f = glob.glob('/tmp/bt/**/s_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def g(r, k):
    return r.get(k) or r.get(k + '_X') or ''
t0 = min(int(r['Start_Timestamp']) for r in rows)
for r in rows:
    n = r['Kernel_Name']
    if 'baq' in n:
        print('%-40s grid %8s wg %4s lds %6s vgpr %4s  start %10.3f ms  dur %8.3f ms' % (n.split('(')[0][-40:], g(r,'Grid_Size'), g(r,'Workgroup_Size'), g(r,'LDS_Block_Size'), g(r,'VGPR_Count'),
              (int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
PY
tail -3 /tmp/bt.log | cut -c1-400 >> $R/gpurun_out/baq_trace.txt
cat $R/gpurun_out/baq_trace.txt
