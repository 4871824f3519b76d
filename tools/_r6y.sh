#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
L=libfx_hip_test.so
for r in 1 2; do for m in 0 4 12 2; do
echo "== FX_SKIP_EMPTY=$m"
FX_SKIP_EMPTY=$m python3 tools/other_configs.py 4 $L 2>&1 | tail -2 | cut -c1-150
done; done
