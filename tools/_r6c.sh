#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { lib=$1; shift; echo -n "$lib $* : "; env "$@" python tools/bench_lib.py $lib $EXTRA 2>&1 | tail -1 | cut -c1-45; }
{
for r in 1 2; do
for l in test noS noR noL; do
run libfx_hip_$l.so FX_FRONT_SPLIT=0
run libfx_hip_$l.so FX_FRONT_SPLIT=1
done
done
echo "--- detector alone (no descriptors), 4 in flight"
EXTRA=--no-descriptors
run libfx_hip_noS.so FX_FRONT_SPLIT=0
run libfx_hip_noS.so FX_FRONT_SPLIT=1
run libfx_hip_noS.so FX_FRONT=0
EXTRA=
run libfx_hip_noS.so FX_FRONT=0
for l in noS; do FX_FRONT_SPLIT=1 python tools/stage_times.py libfx_hip_$l.so 1024 20 | tail -1; done
} > gpurun_out/r6c_bench.log 2>&1
