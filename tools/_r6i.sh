#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl" | tail -8 > gpurun_out/r6i.log
python tools/other_configs.py 0 >> gpurun_out/r6i.log 2>&1
