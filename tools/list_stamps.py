"""Diagnostic: per-phase cycles of the list tier (desc_body in k_desc_mid: rows of 193 .. 1024 support points) on one of
BASELINE's configurations (needs lib/libfx_hip_stamps.so built with -DFX_STAMPS).  python tools/list_stamps.py [2|3|5]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_extraction_amd import capi

capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), "libfx_hip_stamps.so")
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
base = 48
rows = max(v[base + 12], 1)
names = {1: "support set in", 6: "neighbours, cell sort", 7: "density", 2: "bins + weights", 3: "sort by (bin, d2, index)", 4: "bin sums", 5: "tail"}
tot = sum(v[base + k] for k in names)
print(f"{name}: list rows {rows:.0f} (stamped workgroups only), support {v[base + 13] / rows:.0f}, binned neighbours {v[base + 14] / rows:.0f}, {tot / rows:.0f} cycles a row")
for k, nm in names.items():
    print(f"   {nm:28s} {v[base + k] / max(tot, 1) * 100:6.2f} %   {v[base + k] / rows:10.0f}")
