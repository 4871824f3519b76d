#!/usr/bin/env python3
"""After `gpurun -- bash tools/evidence.sh TAG`: condenses gpurun_out/evidence_TAG and gpurun_out/prof_TAG* into the files that are
committed under profiles/ (TAG_*), by running tools/summarize_profile.py TAG, TAG_c1 and tools/summarize_config_profile.py TAG 3|5
and copying the unprofiled runs' lines and logs.
  python tools/collect_evidence.py TAG"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
ev = os.path.join(ROOT, "gpurun_out", "evidence_" + tag)
dst = os.path.join(ROOT, "profiles")
for cmd in (["tools/summarize_profile.py", tag + "_c1"], ["tools/summarize_profile.py", tag], ["tools/summarize_config_profile.py", tag, "3"],
            ["tools/summarize_config_profile.py", tag, "5"]):
    r = subprocess.run([sys.executable] + cmd, cwd=ROOT, capture_output=True, text=True)
    print(" ".join(cmd), "->", r.returncode, (r.stdout + r.stderr)[-600:])
for name in ("bench_default_run.json", "force_dist.json", "bench_one_at_a_time.json", "fx_multi_1gpu.log", "batcher.log", "streaming_latency.jsonl",
             "kernel_info.txt", "fuzz_more.log", "fuzz_dense.log", "fx_multi_selftest8.log",
             "fuzz_more_1000_5000.log", "fuzz_dense_300_1800.log", "determinism.log", "gpu_tests.log", "smoke.log"):  # (the last five: FX_EVIDENCE_LONG=1)
    src = os.path.join(ev, name)
    if os.path.exists(src) and os.path.getsize(src):
        if name.endswith(".log"):  # (without the runtime's noise)
            keep = [l for l in open(src, errors="replace") if "amdgpu.ids" not in l]
            open(os.path.join(dst, f"{tag}_{name}"), "w").writelines(keep[-200:])
        else:
            shutil.copy(src, os.path.join(dst, f"{tag}_{name}"))
        print("copied", name)
    else:
        print("MISSING", name)
