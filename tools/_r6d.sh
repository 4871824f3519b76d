#!/bin/bash
# the full GPU suite + a default bench line on the current tree
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r6d_tests.log
python bench.py --no-extras > gpurun_out/r6d_bench.json 2> gpurun_out/r6d_bench.err
FX_BENCH_FORCE_DIST=1 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r6d_force_dist.json 2> gpurun_out/r6d_force_dist.err
FX_BENCH_FORCE_DIST=1 python bench.py --no-extras --no-cpu-baseline --gather root > gpurun_out/r6d_force_root.json 2> gpurun_out/r6d_force_root.err
