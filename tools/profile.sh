#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + HBM traffic counters + SQ counters for bench.py.
# Counters go in their own passes with --kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE and
# WRITE_SIZE do not fit one pass, 8 SQ slots per pass; never combine --pmc with sys/hip/hsa traces on this pool).
# usage: tools/profile.sh TAG [bench args...]      (FX_PROFILE_SQ=0 skips the two SQ passes)
set -u
TAG=${1:-r01}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# the hardware queues bench.py asks for (its os.environ.setdefault comes too late under the profiler, whose preloaded library
# has started the runtime by then): without them the four contexts' streams share queues and the trace shows another overlap
export GPU_MAX_HW_QUEUES=8
# 100 timed steps, one region (bench.py would repeat it for a second: counters are per launch, one region is enough)
ARGS="--steps 100 --repeats 1 --warmup 2 --no-cpu-baseline --no-extras --check 0 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" $ARGS > "$OUT/bench_trace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$R/bench.py" $ARGS > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$R/bench.py" $ARGS > "$OUT/bench_write.log" 2>&1
if [ "${FX_PROFILE_SQ:-1}" != "0" ]; then
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d "$OUT/sq1" -- python3 "$R/bench.py" $ARGS > "$OUT/bench_sq1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VMEM --output-format csv -d "$OUT/sq2" -- python3 "$R/bench.py" $ARGS > "$OUT/bench_sq2.log" 2>&1
fi
find "$OUT" -name "*.csv" | head -20
grep -o '{"metric.*' "$OUT/bench_trace.log" | head -c 400
