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
n_batches = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 0  # 0: one k_prep launch per batch of the bench workload, counted below
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    if not hits:
        raise SystemExit("missing " + pattern)
    return max(hits, key=os.path.getmtime)  # (gpurun merges runs into the same directory: take the latest)


shutil.copy(one("trace/**/*_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
# the kernel with the largest average launch duration in that summary: the one bench.py's roofline line is about
best = None
for row in csv.DictReader(open(os.path.join(dst, f"{tag}_kernel_stats.csv"))):
    name = row["Name"].split("(")[0]
    if name.startswith("k_") and (best is None or float(row["AverageNs"]) > best[1]):
        best = (name, float(row["AverageNs"]), int(row["Calls"]))
if False:  # (bench.py picks the dominant kernel from its own pre-pass since round 3)
    json.dump({"kernel": best[0], "average_ns": best[1], "calls": best[2],
               "source": f"top AverageNs row among the path's kernels in profiles/{tag}_kernel_stats.csv (rocprofv3 --kernel-trace --stats of bench.py)"},
              open(os.path.join(dst, "dominant.json"), "w"), indent=1)
    print("dominant kernel:", best)


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
FIRST = "k_front" if calls.get("k_front") else "k_prep"  # the kernel launched once per batch
if not n_batches:
    n_batches = calls[FIRST]
print(f"{n_batches} batches in the counter passes")
# ---- does the trace show the run it claims to?
#  (1) the named kernel's average duration in the trace against the SAME run's own measurements of it (bench.py's line in
#      bench_trace.log: kernel_exec_ms = device-clock span of the launch, kernel_ms = HIP events around it): more than 15 %
#      apart means the trace is not of that run, and profiles/traffic.json is NOT written;
#  (2) for information: the sum of all kernels' durations per batch / batches in flight against the run's ms_per_step (they
#      meet only when the chip is never idle: under the profiler every launch costs more and the batches overlap less), and
#      the profiled run's throughput (compare with the unprofiled default run, profiles/TAG_bench_default_run.json).
consistent = True
check_lines = []
try:
    bench_line = [l for l in open(one("bench_trace.log")) if l.startswith('{"metric')][-1]
    bj = json.loads(bench_line)
    stats = list(csv.DictReader(open(os.path.join(dst, f"{tag}_kernel_stats.csv"))))
    first_calls = max(int(r["Calls"]) for r in stats if short(r["Name"]) == FIRST)
    per_batch_ms = sum(float(r["TotalDurationNs"]) for r in stats if short(r["Name"]).startswith("k_")) / first_calls * 1e-6
    k_ctx = bj["config"]["batches_in_flight"]
    rf = bj["roofline"]
    named = rf["kernel"]
    avg_ms = [float(r["AverageNs"]) for r in stats if short(r["Name"]) == named][0] * 1e-6
    own = rf.get("kernel_exec_ms") or rf["kernel_ms"]
    ratio = avg_ms / own
    check_lines.append(f"{named}: rocprofv3 average {avg_ms:.4f} ms; the same run's own clock {rf.get('kernel_exec_ms')} ms (device span), "
                       f"{rf['kernel_ms']:.4f} ms (HIP events, one batch on the chip, after the timed region): ratio to the run's own {ratio:.2f}")
    check_lines.append(f"own algorithmic bytes {rf['alg_bytes_per_launch']:.0f} / rocprofv3 average = {rf['alg_bytes_per_launch'] / (avg_ms * 1e-3) / 1e9:.0f} GB/s "
                       f"= {rf['alg_bytes_per_launch'] / (avg_ms * 1e-3) / 1e9 / 8000:.3f} of the HBM peak (the line of that run says frac {rf['frac']:.3f} = one batch on the chip, "
                       f"frac_exec {rf.get('frac_exec')} = the device span with the other batches in flight)")
    check_lines.append(f"sum of kernel durations per batch {per_batch_ms:.3f} ms / {k_ctx} in flight = {per_batch_ms / k_ctx:.3f} ms; the run's ms_per_step "
                       f"{bj['ms_per_step']:.3f}; the profiled run made {bj['value']:.0f} scans/s")
    consistent = 0.85 <= ratio <= 1.15
    if not consistent:
        check_lines.append("!! the trace's average and the run's own clock are more than 15 % apart: profiles/traffic.json is NOT updated from this run")
    open(os.path.join(dst, f"{tag}_trace_check.txt"), "w").write("\n".join(check_lines) + "\n")
    print("\n".join(check_lines))
except Exception as e:  # (a run without a bench line)
    print("consistency check skipped:", e)
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
if consistent:
    json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
# ---- SQ counters (two passes of eight): where the wave cycles of every kernel go
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sq_summary  # noqa: E402
sq_summary.write_table(src, os.path.join(dst, f"{tag}_sq_counters.csv"), counter, FIRST)
log = one("bench_trace.log")
line = [l for l in open(log) if l.startswith('{"metric')]
if line:
    open(os.path.join(dst, f"{tag}_bench.json"), "w").write(line[-1])
for r in rows:
    print(r)
