"""Diagnostic: per-phase cycles of k_dense_sort on BASELINE config 3 / 5 (needs lib/libfx_hip_stamps.so built with
-DFX_STAMPS).  Usage on the GPU box: python tools/dense_stamps.py [3|5]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_extraction_amd import capi

capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), os.environ.get("FX_STAMPS_LIB", "libfx_hip_stamps.so"))
lib = capi.load()
import bench
import torch

which = sys.argv[1] if len(sys.argv) > 1 else "3"
name = [n for n in bench.OTHER_CONFIGS if n.startswith(f"config{which}")][0]
cfg = bench.OTHER_CONFIGS[name]
B = 32
uniq = [capi.synth_scan(capi.synth_cfg(10 + b, **cfg["synth"])) for b in range(8)]
dev = [torch.from_numpy(s).cuda() for s in uniq]
N = len(uniq[0])
p = capi.params(cfg["preset"], **cfg["params"])
ctx = capi.Context(p, capi.limits(B, N, **dict(cfg["limits"], max_total_keypoints=B * 256)))
descs = ctx.make_descs([dev[b % 8].data_ptr() for b in range(B)], [N] * B, 16, 0.02, -0.015)
for _ in range(2):
    ctx.process_raw(descs, B, capi.FX_IN_DEVICE)
ctx.synchronize()
out = (C.c_ulonglong * 64)()
lib.fx_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
capi.check(lib.fx_debug_stamps(ctx.handle, out))
v = np.array(list(out), dtype=np.float64)
base = 16
rows = max(v[base + 6], 1)
print(f"{name}: k_dense_sort rows {rows:.0f} (stamped workgroups only), support {v[base + 7] / rows:.0f}, overflow region {v[base + 8] / rows:.0f}")
for k, nm in {1: "clear + pass 1 (histogram)", 2: "prefix", 3: "pass 2 (scatter)", 4: "pass 3 (claims, query list)", 5: "items"}.items():
    print(f"   {nm:32s} {v[base + k] / rows:10.0f} cycles per row (100 MHz clock x ?)")
# k_dense_density's diagnostic counters (columns 44..48: free of the ring / merge / descriptor kernels' stamps)
t, d, q, wv, lanes = v[44], v[45], v[46], v[47], v[48]
if q:
    print(f"k_dense_density: {q:.0f} queries in {lanes:.0f} quads ({q / lanes:.2f} per quad); targets walked per quad {t / lanes:.0f}, "
          f"trips per wavefront and 64 quads (unit slots: the longest unit of each) {wv / (lanes / 64):.0f}; true density per query {d / q:.0f}")
# k_dense_finish (LDS path): columns 49..57 of a -DFX_STAMPS -DFX_STAMPS_FINISH build (FX_STAMPS_LIB names it)
if v[57] and "finish" in os.environ.get("FX_STAMPS_LIB", ""):
    n = v[54]
    print(f"k_dense_finish: {n:.0f} rows (stamped workgroups only), binned neighbours {v[55] / n:.0f}, support {v[56] / n:.0f}; cycles per row: "
          f"keys + bins {v[49] / n:.0f}, prefix {v[50] / n:.0f}, order {v[51] / n:.0f}, rank {v[57] / n:.0f}, weights {v[52] / n:.0f}, sorted weights + bin sums {v[53] / n:.0f}")
