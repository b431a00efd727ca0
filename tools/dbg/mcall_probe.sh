#!/bin/bash
# usage (GPU box): bash tools/dbg/mcall_probe.sh  -- the butterfly microtest, the caller's parity tests, the step with and without groups / ploidy
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -o /tmp/rows16 tools/microbench/rows16_product.hip 2>/dev/null && /tmp/rows16 || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mcall_random.py tests/test_gpu_goldens.py tests/test_c_host.py tests/test_gpu_gvcf.py -x -q -m gpu 2>&1 | tail -15 || exit 1
for args in "" "--groups 4 --haploid-frac 0.25" "--groups 4" "--haploid-frac 0.25" "--groups 12"; do
  timeout -k 10 300 python bench.py --extras 0 --cpu-seconds 0 --cpu-all-cores 0 --steps 10 $args 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$args] value %.3g  step %.3f ms  glfgen %.3f  others %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['other_kernels_ms']))"
done
