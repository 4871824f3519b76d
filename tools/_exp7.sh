#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
python tools/bench_lib.py libfx_hip.so 2>&1 | tail -1
python tools/bench_lib.py libfx_hip_sweep2.so 2>&1 | tail -1
done
python tools/bench_lib.py libfx_hip.so --contexts 1 2>&1 | tail -1
python tools/bench_lib.py libfx_hip_sweep2.so --contexts 1 2>&1 | tail -1
