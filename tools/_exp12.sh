#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "$*"; c=$1; shift; env "$@" python tools/bench_lib.py libfx_hip_test.so --contexts $c 2>&1 | tail -1 | cut -c1-45; }
for c in 2 3; do for w in 10 4 2 1; do run $c FX_DESC_WGS_PER_CU=$w; done; done
