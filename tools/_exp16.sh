#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do python tools/bench_lib.py libfx_hip_before.so 2>&1 | tail -1 | cut -c1-45; python tools/bench_lib.py libfx_hip.so 2>&1 | tail -1 | cut -c1-45; done
