#!/bin/bash
cd $GRAFT_REPO_ROOT
for c in 4 3 5 6 4 2; do
echo "contexts $c"; python tools/bench_lib.py libfx_hip.so --contexts $c 2>&1 | tail -1
done
