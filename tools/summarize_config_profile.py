#!/usr/bin/env python3
"""Condenses a tools/profile_config.sh run (rocprofv3 CSVs under gpurun_out/prof_TAG_cfgN) into the files committed
under profiles/:
  profiles/TAG_cfgN_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (verbatim)
  profiles/TAG_cfgN_hbm_traffic.csv    per kernel and batch: duration, FETCH_SIZE, WRITE_SIZE, corrected HBM bytes
  profiles/TAG_cfgN_run.txt            what tools/config_times.py printed under the profiler (batch, flags, stage times)
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, section HBM: both in KiB, FETCH_SIZE half-counts on gfx950).
usage: tools/summarize_config_profile.py TAG 3|5"""
import csv
import glob
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, cfg = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_cfg{cfg}")
dst = os.path.join(ROOT, "profiles")


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    if not hits:
        raise SystemExit("missing " + pattern)
    return max(hits, key=os.path.getmtime)


shutil.copy(one("trace/**/*_kernel_stats.csv"), os.path.join(dst, f"{tag}_cfg{cfg}_kernel_stats.csv"))
with open(os.path.join(dst, f"{tag}_cfg{cfg}_run.txt"), "w") as f:
    f.writelines(l for l in open(os.path.join(src, "trace.log")) if l.startswith(("config", "  ")))


def counter(pattern, cname):
    tot, dur, calls = defaultdict(float), defaultdict(float), defaultdict(int)
    for row in csv.DictReader(open(one(pattern))):
        if row["Counter_Name"] != cname:
            continue
        k = row["Kernel_Name"].split("(")[0]
        tot[k] += float(row["Counter_Value"])
        dur[k] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
        calls[k] += 1
    return tot, dur, calls


if glob.glob(os.path.join(src, "fetch/**/*_counter_collection.csv"), recursive=True):
    fetch, dur, calls = counter("fetch/**/*_counter_collection.csv", "FETCH_SIZE")
    write, _, _ = counter("write/**/*_counter_collection.csv", "WRITE_SIZE")
    nb = calls.get("k_prep") or calls.get("k_prep_sliced") or calls["k_slow"]  # launches per batch: one (k_prep_sliced: small batches of big scans)
    rows = []
    for k in sorted(fetch, key=lambda k: -dur[k]):
        if not k.startswith("k_"):
            continue
        f_kib, w_kib = fetch[k] / nb, write.get(k, 0.0) / nb
        hbm = (2.0 * f_kib + w_kib) * 1024.0
        ms = dur[k] / nb
        rows.append([k, f"{ms:.4f}", f"{f_kib:.1f}", f"{w_kib:.1f}", f"{hbm / 1e6:.3f}", f"{hbm / (ms * 1e-3) / 1e9:.1f}" if ms > 0 else ""])
    with open(os.path.join(dst, f"{tag}_cfg{cfg}_hbm_traffic.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "ms_per_batch(pmc run)", "FETCH_SIZE_KiB_per_batch", "WRITE_SIZE_KiB_per_batch", "hbm_MB_per_batch=(2*FETCH+WRITE)", "hbm_GB_per_s"])
        w.writerows(rows)
    for r in rows[:12]:
        print(r)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sq_summary  # noqa: E402
sq_summary.write_table(src, os.path.join(dst, f"{tag}_cfg{cfg}_sq_counters.csv"), counter, "k_prep" if calls.get("k_prep") else "k_prep_sliced")
print(open(os.path.join(dst, f"{tag}_cfg{cfg}_run.txt")).read())
