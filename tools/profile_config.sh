#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace (+ HBM traffic counter passes) of one of BASELINE.json's
# other configurations (3: 64 x 2048, batch 256; 5: 128 x 2048, R = 2 m, batch 64) through tools/config_times.py.
# usage: tools/profile_config.sh TAG 3|5      (FX_PROFILE_PMC=0 skips the two counter passes)
set -u
TAG=${1:-r03}; CFG=${2:-3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${TAG}_cfg$CFG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/tools/config_times.py" $CFG > "$OUT/trace.log" 2>&1
if [ "${FX_PROFILE_PMC:-1}" != "0" ]; then
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$R/tools/config_times.py" $CFG > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$R/tools/config_times.py" $CFG > "$OUT/write.log" 2>&1
fi
if [ "${FX_PROFILE_SQ:-1}" != "0" ]; then
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d "$OUT/sq1" -- python3 "$R/tools/config_times.py" $CFG > "$OUT/sq1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VMEM --output-format csv -d "$OUT/sq2" -- python3 "$R/tools/config_times.py" $CFG > "$OUT/sq2.log" 2>&1
fi
cat "$OUT/trace.log" | tail -8
