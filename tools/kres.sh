#!/bin/bash
# usage: bash tools/kres.sh <file.hip> [extra flags]  -- registers, scratch and occupancy of every kernel of a source file (compile only, no GPU needed)
cd "$(dirname "$0")/../bcftools_amd/csrc"
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/kres_$$.o 2>&1 |
  grep -E "Function Name|Name:|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* *//; s/ \[-Rpass.*//' |
  awk '/Name:/{if (l) print l; l=$0; next} {l=l" | "$0} END{print l}' | sed 's/Function Name: //; s/  */ /g'
rm -f /tmp/kres_$$.o
