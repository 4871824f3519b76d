#!/bin/bash
# Runs on the GPU box: instruction-cache counters of the headline's kernels (one batch at a time), fused front kernel and
# separate kernels.  usage: tools/icache_profile.sh TAG
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/icache_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d "$OUT/front" -- python3 "$R/tools/bench_lib.py" libfx_hip.so --contexts 1 --steps 20 --repeats 1 > "$OUT/front.log" 2>&1
export FX_FRONT=0
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d "$OUT/separate" -- python3 "$R/tools/bench_lib.py" libfx_hip_test.so --contexts 1 --steps 20 --repeats 1 > "$OUT/separate.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
for mode in ("front", "separate"):
    f = glob.glob(os.path.join(sys.argv[1], mode, "**", "*_counter_collection.csv"), recursive=True)
    if not f:
        print(mode, "no counters"); continue
    tot = defaultdict(lambda: defaultdict(float)); calls = defaultdict(int)
    for row in csv.DictReader(open(max(f, key=os.path.getmtime))):
        k = row["Kernel_Name"].split("(")[0]
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "SQ_WAVES": calls[k] += 1
    print(f"-- {mode}: per launch")
    for k in sorted(tot, key=lambda k: -tot[k]["SQ_WAVE_CYCLES"]):
        if not k.startswith("k_") or not calls[k]: continue
        t = tot[k]; n = calls[k]
        req, hit, miss, dup = (t[x] / n for x in ("SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQC_ICACHE_MISSES_DUPLICATE"))
        print(f"{k:18s} icache req {req:12.0f} hits {hit:12.0f} misses {miss:10.0f} (+dup {dup:10.0f}) miss rate {miss / max(req, 1):.4f}  misses per wave {miss / max(t['SQ_WAVES'] / n, 1):8.1f}  wave Mcycles {t['SQ_WAVE_CYCLES'] * 4 / n / 1e6:8.1f}")
PY
