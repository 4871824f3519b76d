#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + HBM traffic counters for bench.py.
# Counters go in their own passes with --kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE and
# WRITE_SIZE do not fit one pass; never combine --pmc with sys/hip/hsa traces on this pool).
# usage: tools/profile.sh TAG [bench args...]
set -u
TAG=${1:-r01}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-extras --check 0 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" $ARGS > "$OUT/bench_trace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$R/bench.py" $ARGS > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$R/bench.py" $ARGS > "$OUT/bench_write.log" 2>&1
find "$OUT" -name "*.csv" | head -20
grep -o '{"metric.*' "$OUT/bench_trace.log" | head -c 400
