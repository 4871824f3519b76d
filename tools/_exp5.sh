#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_sliced_prep.py tests/test_gpu_configs.py tests/test_gpu_fuzz_dense.py tests/test_gpu_parity.py -x -q 2>&1 | tail -15
python tools/config_times.py 5 2>&1 | tail -4
python tools/config_times.py 3 2>&1 | tail -4
