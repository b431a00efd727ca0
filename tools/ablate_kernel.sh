# usage (on the GPU box): bash tools/ablate_kernel.sh <kernel-name> mask ... [-- bench args]
# Time of one kernel with parts switched off.  The switches exist only in a diagnostics build (-DBCFGPU_DIAG): this
# script rebuilds the library with them, measures, and rebuilds the product library.  16384-site tiles.
k=$1; shift
masks=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do masks+=("$1"); shift; done
[ "$1" == "--" ] && shift
make -s -C bcftools_amd/csrc clean >/dev/null; make -s -j8 -C bcftools_amd/csrc DIAG=1 >/dev/null 2>&1 || { echo "diag build failed"; exit 1; }
for a in "${masks[@]}"; do
  BCFGPU_ABLATE=$a python bench.py --sites 16384 --steps 6 --warmup 2 --cpu-seconds 0 --cpu-all-cores 0 --extras 0 "$@" > gpurun_out/abl_$a.log 2>&1 || { echo fail $a; tail -3 gpurun_out/abl_$a.log; }
  echo "ablate $a: $(grep -o "\"kernel_ms\": [0-9.]*" gpurun_out/abl_$a.log | head -1) $(grep -o "\"$k\": [0-9.]*" gpurun_out/abl_$a.log | head -1)"
done
make -s -C bcftools_amd/csrc clean >/dev/null; make -s -j8 -C bcftools_amd/csrc >/dev/null 2>&1
