#!/bin/bash
cd $GRAFT_REPO_ROOT
for w in 6 3 4 5 2 6 10; do
echo "FX_DESC_WGS_PER_CU=$w"; FX_DESC_WGS_PER_CU=$w python tools/bench_lib.py libfx_hip_test.so 2>&1 | tail -1 | cut -c1-45
done
