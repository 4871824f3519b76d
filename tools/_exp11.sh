#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "$*"; env "$@" python tools/bench_lib.py libfx_hip_test.so 2>&1 | tail -1 | cut -c1-45; }
run FX_DESC_WGS_PER_CU=1
run FX_DESC_WGS_PER_CU=1 FX_GATHER_ROWS=512
run FX_DESC_WGS_PER_CU=1 FX_GATHER_ROWS=256
run FX_DESC_WGS_PER_CU=1 FX_GATHER_ROWS=768
run FX_DESC_WGS_PER_CU=2 FX_GATHER_ROWS=512
run FX_DESC_WGS_PER_CU=1
run FX_DESC_WGS_PER_CU=10
