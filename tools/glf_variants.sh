#!/bin/bash
# glfgen experiments on a GPU box: builds libbcfgpu.so variants (-D flags for glfgen.hip only) and times the headline tile.
# usage: bash tools/glf_variants.sh "<name>:<split 0|1>:<flags>" ...   -> gpurun_out/glfvar.txt ; restores the product build at the end
R=$GRAFT_REPO_ROOT
cd $R/bcftools_amd/csrc
: > $R/gpurun_out/glfvar.txt
for spec in "$@"; do
  name=${spec%%:*}; rest=${spec#*:}; split=${rest%%:*}; flags=${rest#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value $flags -c glfgen.hip -o glfgen.o 2>/dev/null || { echo "$name: build failed" >> $R/gpurun_out/glfvar.txt; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libbcfgpu.so glfgen.o combine.o mcall.o indel.o gap_prep.o baq.o overlap.o pileup.o gvcf.o gather.o capmapq.o api.o tables.o -ldl
  res=$(cd $R && BCFGPU_GLFGEN_SPLIT=$split python3 bench.py --extras 0 --cpu-seconds 0 --cpu-all-cores 0 --steps 8 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f sites/s  step %.3f ms  glfgen %.3f ms  frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))")
  echo "$name (split=$split, $flags): $res" >> $R/gpurun_out/glfvar.txt
done
touch glfgen.hip && make -s
cat $R/gpurun_out/glfvar.txt
