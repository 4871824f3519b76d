"""Diagnostic: stage times and capacity flags of the other BASELINE.json configurations
(3: HDL-64-style 64 x 2048, batch 256; 5: 128 x 2048, R = 2 m, batch 64).  Usage on the GPU box:
  python tools/config_times.py [3|5] [batch]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_extraction_amd import capi

import torch

which = int(sys.argv[1]) if len(sys.argv) > 1 else 3
if len(sys.argv) > 3:
    capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), sys.argv[3])
capi.load()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # the configurations bench.py reports as other_configs

name = [n for n in bench.OTHER_CONFIGS if n.startswith(f"config{which}")][0]
C = bench.OTHER_CONFIGS[name]
B = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) else C["batch"]
cfg = C["synth"]
p = capi.params(C["preset"], **C["params"])
lim = capi.limits(B, cfg["n_rings"] * cfg["n_az"], **dict(C["limits"], max_total_keypoints=B * 256))
# sixteen different scenes, but B DISTINCT device buffers (as bench.py's other_configs leg has them): a batch whose scans alias
# sixteen buffers reads its input from the infinity cache, and the traffic counters then say nothing about HBM
uniq = [capi.synth_scan(capi.synth_cfg(10 + b, **cfg)) for b in range(min(B, 16))]
dev = [torch.from_numpy(uniq[b % len(uniq)]).cuda() for b in range(B)]
ctx = capi.Context(p, lim)
descs = ctx.make_descs([d.data_ptr() for d in dev], [len(uniq[b % len(uniq)]) for b in range(B)], 16, 0.02, -0.015)
steps = 5
ctx.set_profiling(steps)
for _ in range(2):
    v = ctx.process_raw(descs, B, capi.FX_IN_DEVICE | capi.FX_OUT_HOST | capi.FX_OUT_DEBUG)
flags = np.ctypeslib.as_array(v.h_flags, shape=(B,)).copy()
nk = np.ctypeslib.as_array(v.h_n_keypoints, shape=(B,)).copy()
for _ in range(steps):
    ctx.process_raw(descs, B, capi.FX_IN_DEVICE)
ctx.synchronize()
acc, tot = {}, 0.0
for k in range(steps):
    ms, total = ctx.timings(k)
    tot += total
    for n, x in ms.items():
        acc[n] = acc.get(n, 0.0) + x
print(f"config {which}: batch {B}, {len(uniq[0])} points/scan, keypoints/scan {nk.mean():.1f}, flags OR 0x{int(np.bitwise_or.reduce(flags)):x}")
print(f"  total {tot / steps:.3f} ms/batch -> {B / (tot / steps) * 1e3:.0f} scans/s")
print("  " + "  ".join(f"{n}={x / steps:.3f}" for n, x in acc.items()))
import ctypes as C
out = (C.c_uint32 * 16)()
lib = capi.load()
lib.fx_debug_counters.argtypes = [C.c_void_p, C.c_void_p]
capi.check(lib.fx_debug_counters(ctx.handle, out))
names = ["rings -> second run tier", "scans -> big merge", "-", "-", "rows -> list tier",
         "rings -> workgroup tier", "rows -> dense tier", "-", "rows -> wavefront tier", "scans -> huge merge", "-", "-",
         "key-pool entries", "sorted-pool entries", "density work items", "-"]
print("  " + ", ".join(f"{n}: {c}" for n, c in zip(names, out)) + f", rows total {int(nk.sum())}")
nb = np.ctypeslib.as_array(v.h_kp_neighbors, shape=(B, lim.max_keypoints)) if v.h_kp_neighbors else None
if nb is not None:
    allnb = np.concatenate([nb[b, :nk[b]] for b in range(B)])
    print("  neighbours per keypoint: median %d, p90 %d, p99 %d, max %d" % (np.median(allnb), np.percentile(allnb, 90), np.percentile(allnb, 99), allnb.max()))
