#!/bin/bash
# usage (on the GPU box): bash tools/r5_glfgen_fill.sh -- what perfectly filled lanes could give glfgen_kernel: the same tile shape with every cell
# exactly 30 reads deep (no lane of a wavefront waits for a deeper neighbour: the upper bound of any scheme that deals a cell's positions
# out to idle lanes) against Poisson(30); kernel time (HIP events, bench.py) and the SQ counters of both.  -> gpurun_out/r5_glfgen_fill.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUTF=$R/gpurun_out/r5_glfgen_fill.txt
: > $OUTF
for fd in 0 1; do
  python3 $R/bench.py --steps 10 --warmup 2 --cpu-seconds 0 --cpu-all-cores 0 --extras 0 --fixed-depth $fd > $R/gpurun_out/fill_$fd.json 2>/dev/null
  python3 - <<PY >> $OUTF
import json
b=json.loads([l for l in open("$R/gpurun_out/fill_$fd.json") if l.startswith("{")][-1])
print("fixed_depth=$fd: 32768-site tile, reads %d, glfgen_kernel %.3f ms, combine %.3f ms, mcall %.3f ms, step %.3f ms" % (b["config"]["reads_per_tile"], b["roofline"]["kernel_ms"], b["roofline"]["other_kernels_ms"]["combine_kernel"], b["roofline"]["other_kernels_ms"]["mcall_kernel"], b["ms_per_step"]))
PY
  bash $R/tools/pmc_step.sh fill$fd --fixed-depth $fd > /dev/null 2>&1
  echo "  counters (8192-site tile, per dispatch):" >> $OUTF
  grep glfgen $R/gpurun_out/fill${fd}_pmc_step.txt | sed 's/^/    /' >> $OUTF
done
cat $OUTF
