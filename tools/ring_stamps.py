"""Diagnostic: per-phase cycle shares of the ring kernel (needs lib/libfx_hip_stamps.so,
built with -DFX_STAMPS; never the product build).  Usage on the GPU box:
  python tools/ring_stamps.py [preset]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_extraction_amd import capi

capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), os.environ.get("FX_STAMPS_LIB", "libfx_hip_stamps.so"))
lib = capi.load()
preset = sys.argv[1] if len(sys.argv) > 1 else "launch"
B = 256
scans = [capi.synth_scan(capi.synth_cfg(1000 + b)) for b in range(B)]
ctx = capi.Context(capi.params(preset), capi.limits(B, 28800))
descs = ctx.make_descs([s.ctypes.data for s in scans], [len(s) for s in scans], 16, 0.02, -0.015)
for _ in range(3):
    ctx.process_raw(descs, B, 0)
ctx.synchronize()
out = (C.c_ulonglong * 64)()
lib.fx_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
capi.check(lib.fx_debug_stamps(ctx.handle, out))
v = np.array(list(out), dtype=np.float64)
names = {1: "ring split", 2: "run labelling", 3: "all pairs", 4: "find+sizes", 5: "discovery compaction",
         6: "sort replay (1 lane)", 7: "bbox", 8: "gate", 9: "centroid walks", 10: "slot scans", 11: "outputs"}
for tier, base in (("run tier (one wavefront per ring)", 0), ("workgroup tiers", 16), ("merge (per scan x16)", 32)):
    tot = v[base:base + 16].sum()
    print(f"-- {tier}: {tot / (3 * B * 16):.0f} cycles per (scan, ring) slot")
    for k, nm in names.items():
        print(f"   {nm:24s} {v[base + k] / max(tot, 1) * 100:6.2f} %   {v[base + k] / (3 * B * 16):10.0f}")
    cnt = v[base + 12]
    if cnt:
        print(f"   rings {cnt / 3:.0f}/batch  runs/ring {v[base + 13] / cnt:.1f}  segs/ring {v[base + 14] / cnt:.1f}  "
              f"near run pairs/ring {v[base + 15] / cnt:.1f}  near segments scanned/ring {v[base + 0] / cnt:.1f}")

base = 48
tot = v[base:base + 12].sum()
rows = max(v[base + 12], 1)
print(f"-- long-list descriptor tier (fp32 pass): {rows / 3:.0f} rows/batch, support {v[base + 13] / rows:.0f}, neighbours {v[base + 14] / rows:.0f}, "
      f"{tot / rows:.0f} cycles per row")
for k, nm in {1: "load list", 6: "neighbour filter", 7: "density", 2: "bins + weights", 3: "sort", 4: "bin sums", 5: "row write"}.items():
    print(f"   {nm:24s} {v[base + k] / max(tot, 1) * 100:6.2f} %   {v[base + k] / rows:10.0f}")
