#!/bin/bash
# usage (GPU box): bash tools/r4_probe_grp.sh [pmc]  -- where call -G's time goes: step times with groups / ploidy on and off; "pmc": the counters too
cd $GRAFT_REPO_ROOT
for args in "" "--groups 4" "--haploid-frac 0.25" "--groups 4 --haploid-frac 0.25" "--groups 12"; do
  python bench.py --extras 0 --cpu-seconds 0 --cpu-all-cores 0 --steps 10 $args 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$args] value %.3g  step %.3f ms  glfgen %.3f  others %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['other_kernels_ms']))"
done
[ "$1" = pmc ] && bash tools/pmc_step.sh r4grp --groups 4 --haploid-frac 0.25 | grep mcall
