#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | head -5
for rep in 1 2 3; do python tools/bench_lib.py libfx_hip_before.so 2>&1 | tail -1 | cut -c1-45; python tools/bench_lib.py libfx_hip.so 2>&1 | tail -1 | cut -c1-45; done
python3 tools/latency.py 1 300 launch 0 2>/dev/null | tail -2 | head -1 | cut -c100-230
timeout 600 python tools/fuzz_more.py 0 300 2>&1 | tail -1
