#!/bin/bash
# usage (GPU box): bash tools/dbg/glf_pmc_variants.sh "<name>:<flags>" ...  -- SQ counters (tools/pmc_step.sh, 8192-site tile) of the SNP step with glfgen.hip
# built with other options; the variant library is built outside the tree and loaded through BCFGPU_SO.  -> gpurun_out/glf_pmc_variants.txt
R=$GRAFT_REPO_ROOT
ALL="glfgen combine mcall indel gap_prep baq overlap pileup gvcf gather capmapq draw api tables"
PROD=$(make -s -C $R/bcftools_amd/csrc print-flags-glfgen)
OBJS=""; for o in $ALL; do if [ "$o" = glfgen ]; then OBJS="$OBJS /tmp/gpv.o"; else OBJS="$OBJS $R/bcftools_amd/csrc/$o.o"; fi; done
: > $R/gpurun_out/glf_pmc_variants.txt
for spec in "$@"; do
  name=${spec%%:*}; flags="$PROD ${spec#*:}"
  cd $R/bcftools_amd/csrc
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value $flags -c glfgen.hip -o /tmp/gpv.o 2>/dev/null || { echo "$name: build failed"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/gpv.so $OBJS -ldl
  BCFGPU_SO=/tmp/gpv.so bash $R/tools/pmc_step.sh gpv_$name > /dev/null 2>&1
  echo "== $name ($flags)" >> $R/gpurun_out/glf_pmc_variants.txt
  grep "glfgen_kernel<false, true, false, 0>" $R/gpurun_out/gpv_${name}_pmc_step.txt >> $R/gpurun_out/glf_pmc_variants.txt
done
echo "== the other kernels of the step (product build)" >> $R/gpurun_out/glf_pmc_variants.txt
grep -v glfgen $R/gpurun_out/gpv_${1%%:*}_pmc_step.txt >> $R/gpurun_out/glf_pmc_variants.txt
cat $R/gpurun_out/glf_pmc_variants.txt
