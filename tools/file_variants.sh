#!/bin/bash
# Experiments on a GPU box: builds libbcfgpu.so variants (-D flags for ONE source file), runs a bench mode under rocprofv3 --stats
# and keeps the average times of the kernels whose names match.
# usage: bash tools/file_variants.sh <file.hip> <kernel-name-pattern> "<bench args>" "<name>:<flags>" ...
# -> gpurun_out/filevar.txt ; restores the product build at the end
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
SRC=$1; PAT=$2; BARGS=$3; shift 3
: > $R/gpurun_out/filevar.txt
OBJS="glfgen.o combine.o mcall.o indel.o gap_prep.o baq.o overlap.o pileup.o gvcf.o gather.o capmapq.o draw.o api.o tables.o"
PROD=$(make -s -C $R/bcftools_amd/csrc print-flags-${SRC%.hip})          # the product's own options of this file (csrc/Makefile)
for spec in "$@"; do
  name=${spec%%:*}; flags="$PROD ${spec#*:}"
  cd $R/bcftools_amd/csrc
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value $flags -c $SRC -o ${SRC%.hip}.o 2>/dev/null || { echo "$name: build failed" >> $R/gpurun_out/filevar.txt; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libbcfgpu.so $OBJS -ldl
  cd /tmp; rm -rf /tmp/fv
  rocprofv3 --kernel-trace --stats -d /tmp/fv -o s --output-format csv -- python3 $R/bench.py $BARGS > /tmp/fv.log 2>&1 || { echo "$name: run failed" >> $R/gpurun_out/filevar.txt; tail -3 /tmp/fv.log; continue; }
  k=$(python3 -c "
import csv
for r in csv.DictReader(open('/tmp/fv/s_kernel_stats.csv')):
    n=r['Name']
    if '$PAT' in n: print(n.split('(')[0].replace('void ','').replace('bcfgpu::',''), 'calls', r['Calls'], 'total %.2f ms avg %.3f max %.3f;' % (float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e6, float(r['MaxNs'])/1e6), end=' ')
")
  echo "$name ($flags): $k" >> $R/gpurun_out/filevar.txt
done
cd $R/bcftools_amd/csrc && touch $SRC && make -s
cat $R/gpurun_out/filevar.txt
