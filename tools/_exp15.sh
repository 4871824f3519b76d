#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | head -5
for rep in 1 2; do python tools/bench_lib.py libfx_hip.so 2>&1 | tail -1 | cut -c1-45; done
python tools/other_configs.py 4 2>&1 | grep config | cut -c1-130
timeout 600 python tools/fuzz_more.py 0 300 2>&1 | tail -1
