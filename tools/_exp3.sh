#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fuzz_dense.py tests/test_gpu_configs.py tests/test_gpu_row_reuse_dense.py tests/test_gpu_row_reuse.py tests/test_gpu_front.py -x -q 2>&1 | tail -15
for rep in 1 2; do
python tools/bench_lib.py libfx_hip.so 2>&1 | tail -1
FX_DENSE_SLOW=0 python tools/bench_lib.py libfx_hip_test.so 2>&1 | tail -1
FX_DENSE_SLOW=1 python tools/bench_lib.py libfx_hip_test.so 2>&1 | tail -1
done
