#!/bin/bash
# Experiments on a GPU box: builds variants of libbcfgpu.so (-D flags for ONE source file) OUTSIDE the tree (/tmp), loads each in place
# of the product library (BCFGPU_SO, bcftools_amd/lib.py) for a bench mode under rocprofv3 --stats and keeps the average times of the
# kernels whose names match.  Neither the product library nor its object files are touched.
# usage: bash tools/file_variants.sh <file.hip> <kernel-name-pattern> "<bench args>" "<name>:<flags>" ...
# -> gpurun_out/filevar.txt
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
SRC=$1; PAT=$2; BARGS=$3; shift 3
: > $R/gpurun_out/filevar.txt
ALL="glfgen combine mcall indel gap_prep baq overlap pileup gvcf gather capmapq draw api tables"
PROD=$(make -s -C $R/bcftools_amd/csrc print-flags-${SRC%.hip})          # the product's own options of this file (csrc/Makefile)
OBJS=""
for o in $ALL; do if [ "$o" = "${SRC%.hip}" ]; then OBJS="$OBJS /tmp/fv_variant.o"; else OBJS="$OBJS $R/bcftools_amd/csrc/$o.o"; fi; done
trap 'rm -f /tmp/fv_variant.o /tmp/fv_variant.so' EXIT
for spec in "$@"; do
  name=${spec%%:*}; flags="$PROD ${spec#*:}"
  cd $R/bcftools_amd/csrc
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value $flags -c $SRC -o /tmp/fv_variant.o 2>/dev/null || { echo "$name: build failed" >> $R/gpurun_out/filevar.txt; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/fv_variant.so $OBJS -ldl
  cd /tmp; rm -rf /tmp/fv
  BCFGPU_SO=/tmp/fv_variant.so rocprofv3 --kernel-trace --stats -d /tmp/fv -o s --output-format csv -- python3 $R/bench.py $BARGS > /tmp/fv.log 2>&1 || { echo "$name: run failed" >> $R/gpurun_out/filevar.txt; tail -3 /tmp/fv.log; continue; }
  k=$(python3 -c "
import csv
for r in csv.DictReader(open('/tmp/fv/s_kernel_stats.csv')):
    n=r['Name']
    if '$PAT' in n: print(n.split('(')[0].replace('void ','').replace('bcfgpu::',''), 'calls', r['Calls'], 'total %.2f ms avg %.3f max %.3f;' % (float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e6, float(r['MaxNs'])/1e6), end=' ')
")
  echo "$name ($flags): $k" >> $R/gpurun_out/filevar.txt
done
cat $R/gpurun_out/filevar.txt
