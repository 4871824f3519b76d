#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for m in 0 1 2 3; do
  echo "== FX_SKIP_EMPTY=$m"
  FX_SKIP_EMPTY=$m timeout 300 python tools/bench_lib.py libfx_hip_test.so 2>&1 | tail -1
done
done
