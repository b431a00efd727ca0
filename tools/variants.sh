#!/bin/bash
# usage (on the GPU box): bash tools/variants.sh "<flags1>" "<flags2>" ...   -- rebuild glfgen.hip with extra compiler flags
# (e.g. "-DGLF_WAVES=5 -DFU=2"; VAR_ARGS="--depth 10" adds bench arguments), relink and time the kernel on a 16384-site tile; restores the default build at the end.
cd $GRAFT_REPO_ROOT/bcftools_amd/csrc
for f in "$@"; do
  src=glfgen.hip
  case "$f" in PREV*) src=glfgen_prev.hip; f="${f#PREV}";; esac       # "PREV <flags>": an older copy of the kernel kept beside it, as the in-run reference
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $f -c $src -o glfgen.o 2>/dev/null && make -s ../libbcfgpu.so >/dev/null 2>&1
  r=$(cd $GRAFT_REPO_ROOT && python bench.py --sites 16384 --steps 8 --warmup 2 --cpu-seconds 0 --cpu-all-cores 0 --extras 0 $VAR_ARGS 2>/dev/null | grep -o "kernel_ms.: [0-9.]*")
  echo "[$f] $r"
done
rm -f glfgen.o; make -s ../libbcfgpu.so >/dev/null 2>&1
