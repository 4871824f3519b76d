"""Diagnostic: per-stage device ms of the bench workload for an alternative build of the library
(e.g. an experiment compiled with extra -D flags into lib/<name>).  Usage on the GPU box:
  python tools/stage_times.py [lib file name under feature_extraction_amd/lib] [batch] [steps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_extraction_amd import capi

if len(sys.argv) > 1 and sys.argv[1] != "-":
    capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
import torch

capi.load()
scans = [capi.synth_scan(capi.synth_cfg(1000 + b)) for b in range(B)]
dev = [torch.from_numpy(s).cuda() for s in scans]
over = {}
if os.environ.get("FX_MAX_NEIGHBORS"):
    over["max_neighbors"] = int(os.environ["FX_MAX_NEIGHBORS"])
ctx = capi.Context(capi.params(os.environ.get("FX_PRESET", "launch")), capi.limits(B, 28800, **over))
descs = ctx.make_descs([d.data_ptr() for d in dev], [len(s) for s in scans], 16, 0.02, -0.015)
ctx.set_profiling(steps)
for _ in range(3):
    ctx.process_raw(descs, B, capi.FX_IN_DEVICE)
ctx.synchronize()
for _ in range(steps):
    ctx.process_raw(descs, B, capi.FX_IN_DEVICE)
ctx.synchronize()
acc, tot = {}, 0.0
for k in range(steps):
    ms, total = ctx.timings(k)
    tot += total
    for n, v in ms.items():
        acc[n] = acc.get(n, 0.0) + v
print(f"{os.path.basename(capi.LIB_PATH)}: total {tot / steps:.4f} ms/batch -> {B / (tot / steps) * 1e3:.0f} scans/s")
print("  " + "  ".join(f"{n}={v / steps:.3f}" for n, v in acc.items()))
