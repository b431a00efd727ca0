#!/bin/bash
# usage (here, not on the GPU box): bash tools/gpu.sh [--timeout N] -- '<command>'   -- rebuild every in-tree binary, then gpurun
# (the snapshot carries the built .so files: a stale one is what the GPU box would test)
cd "$(dirname "$0")/.." || exit 1
python -c "import __graft_entry__ as g; g.build()" || { echo "build failed"; exit 1; }
exec /usr/local/graft/bin/gpurun "$@"
