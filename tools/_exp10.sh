#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "$*"; env "$@" python tools/bench_lib.py libfx_hip_test.so --contexts 1 2>&1 | tail -1 | cut -c1-45; }
for w in 10 4 2 1; do run FX_DESC_WGS_PER_CU=$w; done
for w in 10 4 2 1; do echo "cfg3/5 w=$w"; FX_DESC_WGS_PER_CU=$w python tools/other_configs.py 4 libfx_hip_test.so 2>&1 | grep config | cut -c1-120; done
