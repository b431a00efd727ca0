#!/bin/bash
# usage (on the GPU box): bash tools/variants.sh "<flags1>" "<flags2>" ...   -- rebuild glfgen.hip with extra compiler flags
# (e.g. "-DGLF_WAVES=5 -DFU=2"; VAR_ARGS="--depth 10" adds bench arguments), relink and time the kernel on a 16384-site tile; restores the default build at the end.
cd $GRAFT_REPO_ROOT/bcftools_amd/csrc
for f in "$@"; do
  src=${VAR_SRC:-glfgen.hip}; obj=${src%.hip}.o                        # VAR_SRC=combine.hip: another kernel's source
  case "$f" in PREV*) src=glfgen_prev.hip; obj=glfgen.o; f="${f#PREV}";; esac       # "PREV <flags>": an older copy of the kernel kept beside it, as the in-run reference
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $f -c $src -o $obj 2>/dev/null && make -s ../libbcfgpu.so >/dev/null 2>&1
  r=$(cd $GRAFT_REPO_ROOT && python bench.py --sites 16384 --steps 8 --warmup 2 --cpu-seconds 0 --cpu-all-cores 0 --extras 0 $VAR_ARGS 2>/dev/null | grep -o "kernel_ms.: [0-9.]*\|other_kernels_ms.: {[^}]*}" | tr '\n' ' ')
  echo "[$f] $r"
done
rm -f glfgen.o ${VAR_SRC%.hip}.o; make -s ../libbcfgpu.so >/dev/null 2>&1
