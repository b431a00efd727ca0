#!/bin/bash
# usage (GPU box): bash tools/dbg/glf_probe.sh  -- parity tests of the mpileup stage, then glfgen.hip variants timed under rocprofv3 on the headline tile
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_goldens.py tests/test_gpu_draw.py -x -q -m gpu 2>&1 | tail -15 || exit 1
bash tools/file_variants.sh glfgen.hip glfgen_kernel "--extras 0 --cpu-seconds 0 --cpu-all-cores 0 --steps 10" "$@" 2>&1 | grep -v "^$" | cut -c1-330
