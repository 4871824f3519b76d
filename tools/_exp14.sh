#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "$*"; env "$@" python tools/bench_lib.py libfx_hip_test.so 2>&1 | tail -1 | cut -c1-45; }
for rep in 1 2 3; do run FX_TIER_MIN_GRID=8; run FX_TIER_MIN_GRID=1; run FX_TIER_MIN_GRID=2; done
