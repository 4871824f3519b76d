#!/bin/bash
# Runs on the GPU box (through gpurun): everything the round's numbers in DESIGN.md / README.md are quoted from, in one call.
# usage: tools/evidence.sh TAG        -> gpurun_out/evidence_TAG/* and gpurun_out/prof_TAG*/ (summarised afterwards with
#        tools/summarize_profile.py TAG, TAG_c1 and tools/summarize_config_profile.py TAG 3|5)
set -u
TAG=${1:-r06d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
E=$R/gpurun_out/evidence_$TAG
mkdir -p "$E"
cd "$R"
bash tools/kernel_info.sh > "$E/kernel_info.txt" 2>&1   # registers, spills, scratch, LDS and code size of every kernel of the product library
python3 bench.py > "$E/bench_default_run.json" 2> "$E/bench_default_run.err"
FX_BENCH_FORCE_DIST=1 python3 bench.py --no-extras --no-cpu-baseline > "$E/force_dist.json" 2> "$E/force_dist.err"
python3 bench.py --contexts 1 --no-extras --no-cpu-baseline > "$E/bench_one_at_a_time.json" 2> /dev/null
feature_extraction_amd/bin/fx_multi_cli --batch 1024 --steps 40 --inflight 1 > "$E/fx_multi_1gpu.log" 2>&1
feature_extraction_amd/bin/fx_multi_cli --batch 1024 --steps 40 --inflight 2 >> "$E/fx_multi_1gpu.log" 2>&1
feature_extraction_amd/bin/fx_multi_cli --batch 1024 --steps 40 --inflight 4 >> "$E/fx_multi_1gpu.log" 2>&1
feature_extraction_amd/bin/fx_multi_cli --batch 1024 --steps 20 --inflight 4 --host-input >> "$E/fx_multi_1gpu.log" 2>&1
feature_extraction_amd/bin/fx_multi_cli --batch 1024 --steps 40 --inflight 4 --root 0 >> "$E/fx_multi_1gpu.log" 2>&1
feature_extraction_amd/bin/fx_batcher_cli --sensors 4 --hz 10 --seconds 3 --out /tmp/batcher_a.bin > "$E/batcher.log" 2>&1
feature_extraction_amd/bin/fx_batcher_cli --sensors 8 --burst 32 --out /tmp/batcher_b.bin >> "$E/batcher.log" 2>&1
: > "$E/streaming_latency.jsonl"
for b in 1 8 16 64; do python3 tools/latency.py $b 200 launch 1 2>/dev/null | tail -2 >> "$E/streaming_latency.jsonl"; done
python3 tools/latency.py 1 200 launch 0 2>/dev/null | tail -2 >> "$E/streaming_latency.jsonl"
bash tools/profile.sh $TAG > "$E/profile.log" 2>&1
bash tools/profile.sh ${TAG}_c1 --contexts 1 > "$E/profile_c1.log" 2>&1
FX_PROFILE_PMC=1 bash tools/profile_config.sh $TAG 3 > "$E/profile_cfg3.log" 2>&1
FX_PROFILE_PMC=1 bash tools/profile_config.sh $TAG 5 > "$E/profile_cfg5.log" 2>&1
# parity beyond the suite, on this binary: the VLP-16 fuzz through every front path (eight since round 6), the dense fuzz, determinism
timeout 2700 python3 tools/fuzz_more.py 0 1000 > "$E/fuzz_more.log" 2>&1
timeout 900 python3 tools/fuzz_dense.py 0 300 > "$E/fuzz_dense.log" 2>&1
feature_extraction_amd/bin/fx_multi_cli --selftest 8 --batch 61 --steps 6 --inflight 3 --bad-scan 60 > "$E/fx_multi_selftest8.log" 2>&1
# FX_EVIDENCE_LONG=1: 4000 more fuzz seeds x 9 paths, 1500 more dense seeds, 30 repetitions of a 64-scan batch, the GPU suite, smoke()
if [ "${FX_EVIDENCE_LONG:-0}" = "1" ]; then
  timeout 1500 python3 tools/fuzz_more.py 1000 5000 > "$E/fuzz_more_1000_5000.log" 2>&1
  timeout 300 python3 tools/fuzz_dense.py 300 1800 > "$E/fuzz_dense_300_1800.log" 2>&1
  timeout 300 python3 tools/stress_determinism.py > "$E/determinism.log" 2>&1
  timeout 1800 python3 -m pytest tests -m gpu -q > "$E/gpu_tests.log" 2>&1
  python3 -c "import __graft_entry__ as g; g.smoke()" > "$E/smoke.log" 2>&1
fi
ls "$E"
