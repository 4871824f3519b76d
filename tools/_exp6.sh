#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for rep in 1 2; do
python tools/bench_lib.py libfx_hip_before.so 2>&1 | tail -1
python tools/bench_lib.py libfx_hip.so 2>&1 | tail -1
done
python tools/config_times.py 5 2>&1 | grep -A1 total
python tools/config_times.py 3 2>&1 | grep -A1 total
timeout 600 python tools/fuzz_more.py 0 600 2>&1 | tail -1
timeout 600 python tools/fuzz_dense.py 0 200 2>&1 | tail -2
