# usage: bash tools/ablate_kernel.sh <kernel-name> mask ... [-- bench args]   -- time of one kernel with parts switched off (BCFGPU_ABLATE);
# 16384-site tiles, the size DESIGN.md quotes the per-part times for (a later --sites in the bench args overrides it)
k=$1; shift
masks=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do masks+=("$1"); shift; done
[ "$1" == "--" ] && shift
for a in "${masks[@]}"; do
  BCFGPU_ABLATE=$a python bench.py --sites 16384 --steps 6 --warmup 2 --cpu-seconds 0 --cpu-all-cores 0 "$@" > gpurun_out/abl_$a.log 2>&1 || { echo fail $a; tail -3 gpurun_out/abl_$a.log; }
  echo "ablate $a: $(grep -o "\"$k\": [0-9.]*" gpurun_out/abl_$a.log | head -1)"
done
