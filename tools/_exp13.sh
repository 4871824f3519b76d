#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_multi.py tests/test_gpu_bench_multi.py -x -q 2>&1 | tail -3
for rep in 1 2; do python tools/bench_lib.py libfx_hip.so 2>&1 | tail -1 | cut -c1-45; done
python tools/bench_lib.py libfx_hip.so --contexts 1 2>&1 | tail -1 | cut -c1-45
python tools/bench_lib.py libfx_hip.so --contexts 3 2>&1 | tail -1 | cut -c1-45
feature_extraction_amd/bin/fx_multi_cli --batch 1024 --steps 40 --inflight 4 2>&1 | tail -1
