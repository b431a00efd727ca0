#!/bin/bash
# usage (GPU box): bash tools/dbg/snp_probe.sh  -- the mpileup-stage parity tests, then the headline step three times
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_goldens.py -x -q -m gpu 2>&1 | tail -3 || exit 1
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --extras 0 --cpu-seconds 0 --cpu-all-cores 0 --steps 10 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('value %.4g  step %.3f ms  glfgen %.3f  others %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['other_kernels_ms']))"
done
