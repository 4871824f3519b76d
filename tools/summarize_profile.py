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
# ---- SQ counters (two passes of eight): where the wave cycles of every kernel go
sq_names = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
            "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM",
            "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVES", "SQ_ACTIVE_INST_VMEM"]
if glob.glob(os.path.join(src, "sq1/**/*_counter_collection.csv"), recursive=True):
    sq, sq_dur = {}, {}
    for i, cname in enumerate(sq_names):
        tot, d, cl = counter(("sq1" if i < 8 else "sq2") + "/**/*_counter_collection.csv", cname)
        sq[cname] = tot
        if cname == "SQ_WAVE_CYCLES":
            sq_dur, nb_sq = d, cl["k_prep"]
    with open(os.path.join(dst, f"{tag}_sq_counters.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "ms_per_batch(pmc run)", "waves_per_batch", "wave_Mcycles_per_batch(quad-cycles x4)",
                    "frac_wait_any(s_waitcnt/barrier)", "frac_wait_inst_any(issue stall)", "frac_active_inst_any",
                    "frac_active_valu", "frac_active_lds", "frac_wait_inst_lds", "frac_active_vmem",
                    "mean_waves_in_flight(of 8192 slots)", "valu_insts_per_wave", "lds_insts_per_wave", "salu_insts_per_wave",
                    "vmem_insts_per_wave", "lds_bank_conflict_frac_of_lds_active"])
        for k in sorted(sq_dur, key=lambda k: -sq_dur[k]):
            if not k.startswith("k_"):
                continue
            wc = sq["SQ_WAVE_CYCLES"][k]
            if wc <= 0:
                continue
            waves = max(sq["SQ_WAVES"].get(k, 0.0), 1.0)
            ms = sq_dur[k] / nb_sq
            # SQ_WAVE_CYCLES and the WAIT / ACTIVE counters are in quad-cycles (MI355X_MICROARCH.md, cycle constants);
            # mean waves in flight = wave cycles / kernel cycles, with the kernel's cycles from its duration at 2.4 GHz
            in_flight = (wc * 4.0 / nb_sq) / (ms * 1e-3 * 2.4e9) if ms > 0 else 0.0
            fr = lambda n: f"{sq[n].get(k, 0.0) / wc:.3f}"  # noqa: E731
            lds_act = sq["SQ_LDS_IDX_ACTIVE"].get(k, 0.0)
            w.writerow([k, f"{ms:.4f}", f"{waves / nb_sq:.0f}", f"{wc * 4.0 / nb_sq / 1e6:.2f}", fr("SQ_WAIT_ANY"), fr("SQ_WAIT_INST_ANY"),
                        fr("SQ_ACTIVE_INST_ANY"), fr("SQ_ACTIVE_INST_VALU"), fr("SQ_ACTIVE_INST_LDS"), fr("SQ_WAIT_INST_LDS"),
                        fr("SQ_ACTIVE_INST_VMEM"), f"{in_flight:.0f}",
                        f"{sq['SQ_INSTS_VALU'].get(k, 0.0) / waves:.0f}", f"{sq['SQ_INSTS_LDS'].get(k, 0.0) / waves:.0f}",
                        f"{sq['SQ_INSTS_SALU'].get(k, 0.0) / waves:.0f}", f"{sq['SQ_INSTS_VMEM'].get(k, 0.0) / waves:.0f}",
                        f"{sq['SQ_LDS_BANK_CONFLICT'].get(k, 0.0) / lds_act:.3f}" if lds_act > 0 else ""])
    print(open(os.path.join(dst, f"{tag}_sq_counters.csv")).read())
log = one("bench_trace.log")
line = [l for l in open(log) if l.startswith('{"metric')]
if line:
    open(os.path.join(dst, f"{tag}_bench.json"), "w").write(line[-1])
for r in rows:
    print(r)
