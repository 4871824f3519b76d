#!/usr/bin/env python3
"""Condenses a tools/profile.sh run (rocprofv3 CSVs under gpurun_out/prof_TAG) into the small
files that are committed under profiles/:
  profiles/TAG_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (verbatim)
  profiles/TAG_hbm_traffic.csv    per kernel and batch: duration, FETCH_SIZE, WRITE_SIZE, corrected HBM bytes
  profiles/traffic.json           corrected HBM bytes per batch per kernel (bench.py's roofline.traffic)
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB and on gfx950 FETCH_SIZE
reports exactly half of a wide coalesced read (MI355X_MICROARCH.md, section HBM).
usage: tools/summarize_profile.py TAG [n_batches]
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
n_batches = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # 0: one k_prep launch per batch of the bench workload, counted below
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    if not hits:
        raise SystemExit("missing " + pattern)
    return hits[0]


shutil.copy(one("trace/**/*_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))


def short(name):
    return name.split("(")[0]


def counter(pattern, cname):
    tot, dur, calls = defaultdict(float), defaultdict(float), defaultdict(int)
    for row in csv.DictReader(open(one(pattern))):
        if row["Counter_Name"] != cname:
            continue
        k = short(row["Kernel_Name"])
        tot[k] += float(row["Counter_Value"])
        dur[k] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
        calls[k] += 1
    return tot, dur, calls


fetch, dur_f, calls = counter("fetch/**/*_counter_collection.csv", "FETCH_SIZE")
if not n_batches:
    n_batches = calls["k_prep"]
print(f"{n_batches} batches in the counter passes")
write, _, _ = counter("write/**/*_counter_collection.csv", "WRITE_SIZE")
rows, traffic = [], {}
for k in sorted(fetch, key=lambda k: -dur_f[k]):
    if not k.startswith("k_"):
        continue
    f_kib, w_kib = fetch[k] / n_batches, write.get(k, 0.0) / n_batches
    hbm = (2.0 * f_kib + w_kib) * 1024.0
    ms = dur_f[k] / n_batches
    rows.append([k, calls[k] / n_batches, f"{ms:.4f}", f"{f_kib:.1f}", f"{w_kib:.1f}", f"{hbm / 1e6:.3f}",
                 f"{hbm / (ms * 1e-3) / 1e9:.1f}" if ms > 0 else ""])
    traffic[k] = hbm
with open(os.path.join(dst, f"{tag}_hbm_traffic.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches_per_batch", "ms_per_batch(pmc run)", "FETCH_SIZE_KiB_per_batch", "WRITE_SIZE_KiB_per_batch",
                "hbm_MB_per_batch=(2*FETCH+WRITE)", "hbm_GB_per_s"])
    w.writerows(rows)
traffic["_source"] = f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), run {tag}; bytes per batch of the bench workload"
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
log = one("bench_trace.log")
line = [l for l in open(log) if l.startswith('{"metric')]
if line:
    open(os.path.join(dst, f"{tag}_bench.json"), "w").write(line[-1])
for r in rows:
    print(r)
