#!/bin/bash
# pileup experiments on a GPU box: builds libbcfgpu.so variants (-D flags for pileup.hip only), runs bench.py --mode pileup under
# rocprofv3 --stats and keeps the pileup kernels' average times.  usage: bash tools/pileup_variants.sh "<name>:<flags>" ...
# -> gpurun_out/pileupvar.txt ; restores the product build at the end
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
: > $R/gpurun_out/pileupvar.txt
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  cd $R/bcftools_amd/csrc
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value $flags -c pileup.hip -o pileup.o 2>/dev/null || { echo "$name: build failed" >> $R/gpurun_out/pileupvar.txt; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libbcfgpu.so glfgen.o combine.o mcall.o indel.o gap_prep.o baq.o overlap.o pileup.o gvcf.o gather.o capmapq.o api.o tables.o -ldl
  cd /tmp; rm -rf /tmp/pv
  rocprofv3 --kernel-trace --stats -d /tmp/pv -o s --output-format csv -- python3 $R/bench.py --mode pileup --steps 4 > /tmp/pv.log 2>&1 || { echo "$name: run failed" >> $R/gpurun_out/pileupvar.txt; tail -3 /tmp/pv.log; continue; }
  k=$(python3 -c "
import csv
for r in csv.DictReader(open('/tmp/pv/s_kernel_stats.csv')):
    n=r['Name']
    if 'pileup' in n or 'unpack' in n: print(n.split('(')[0].replace('void ','').replace('bcfgpu::',''), '%.3f ms;' % (float(r['AverageNs'])/1e6), end=' ')
")
  b=$(grep '^{' /tmp/pv.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); h=d['host_fed_pipeline']; print('call %.1f ms, overlapped %.1f ms/region = %.0f sites/s' % (d['whole_call_ms'], h['overlapped_ms_per_region'], h['sites_per_s']))")
  echo "$name ($flags): $k $b" >> $R/gpurun_out/pileupvar.txt
done
cd $R/bcftools_amd/csrc && touch pileup.hip && make -s
cat $R/gpurun_out/pileupvar.txt
