#!/bin/bash
# Runs on the GPU box: SQ counter passes (two of eight counters) of tools/config_times.py for one configuration.
# usage: tools/profile_config_sq.sh TAG 3|5
set -u
TAG=${1:-r03}; CFG=${2:-3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${TAG}_cfg$CFG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d "$OUT/sq1" -- python3 "$R/tools/config_times.py" $CFG > "$OUT/sq1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VMEM --output-format csv -d "$OUT/sq2" -- python3 "$R/tools/config_times.py" $CFG > "$OUT/sq2.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for d in ("sq1", "sq2"):
    for f in glob.glob(f"{out}/{d}/**/*_counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            if row["Counter_Name"] in ("SQ_WAVE_CYCLES",):
                acc[k]["_ns"] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
                acc[k]["_calls"] += 1
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("_ns", 0))[:10]:
    wc = max(c.get("SQ_WAVE_CYCLES", 0), 1); waves = max(c.get("SQ_WAVES", 0), 1); calls = max(c.get("_calls", 0), 1)
    print(f"{k:20s} {c['_ns'] / calls / 1e6:7.3f} ms  wait_any {c['SQ_WAIT_ANY'] / wc:.2f} wait_inst {c['SQ_WAIT_INST_ANY'] / wc:.2f} active_any {c['SQ_ACTIVE_INST_ANY'] / wc:.2f} "
          f"valu {c['SQ_ACTIVE_INST_VALU'] / wc:.2f} lds {c['SQ_ACTIVE_INST_LDS'] / wc:.2f} wait_lds {c['SQ_WAIT_INST_LDS'] / wc:.2f} | per wave: valu {c['SQ_INSTS_VALU'] / waves:.0f} lds {c['SQ_INSTS_LDS'] / waves:.0f} "
          f"salu {c['SQ_INSTS_SALU'] / waves:.0f} vmem {c['SQ_INSTS_VMEM'] / waves:.0f} | bank conflict {c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1):.2f} waves/call {waves / calls:.0f}")
PY
