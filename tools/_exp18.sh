#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "$*"; env "$@" python tools/bench_lib.py libfx_hip_test.so 2>&1 | tail -1 | cut -c1-45; }
run FX_DESC_GRID=256
run FX_DESC_GRID=128
run FX_DESC_GRID=64
run FX_DESC_GRID=192
run FX_DESC_GRID=384
run FX_DESC_GRID=256
