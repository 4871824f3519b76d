"""Diagnostic: per-phase cycle shares of k_front (needs lib/libfx_hip_stamps.so, built with -DFX_STAMPS; never the
product build).  Usage on the GPU box:  python tools/front_stamps.py [preset] [batch]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_extraction_amd import capi

capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), os.environ.get("FX_STAMPS_LIB", "libfx_hip_stamps.so"))
lib = capi.load()
preset = sys.argv[1] if len(sys.argv) > 1 else "launch"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
scans = [capi.synth_scan(capi.synth_cfg(1000 + b)) for b in range(min(B, 64))]
scans = [scans[b % len(scans)] for b in range(B)]
ctx = capi.Context(capi.params(preset), capi.limits(B, 28800))
descs = ctx.make_descs([s.ctypes.data for s in scans], [len(s) for s in scans], 16, 0.02, -0.015)
REP = 3
for _ in range(REP):
    ctx.process_raw(descs, B, 0)
ctx.synchronize()
out = (C.c_ulonglong * 64)()
lib.fx_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
capi.check(lib.fx_debug_stamps(ctx.handle, out))
v = np.array(list(out), dtype=np.float64)
names = {1: "A streaming pass", 2: "B ring split", 3: "C run labelling (masks, prefix)", 4: "C tables (segments, boxes)", 5: "C near run pairs",
         6: "C cross-run edges", 7: "C roots + sizes", 8: "C cluster enumeration", 9: "C sort partition", 10: "C ranking", 11: "C box fold + gate",
         12: "C centroids", 13: "C slots", 14: "C keypoint_cloud", 15: "D merge"}
tot = sum(v[k] for k in names)
print(f"k_front: {tot / (REP * B):.0f} clock ticks per scan (s_memtime)")
for k, nm in names.items():
    print(f"   {nm:34s} {v[k] / max(tot, 1) * 100:6.2f} %   {v[k] / (REP * B):10.0f}")
print("   A detail: setup+first loads", v[24] / (REP * B), "load wait + rotate + tests", v[25] / (REP * B), "barrier", v[26] / (REP * B), "compaction", v[27] / (REP * B), "sweeps", v[28] / (REP * B), "copies (next tile's load wait)", v[29] / (REP * B))
print("   A detail 2: issue next tile's loads", v[30] / (REP * B), "(the rest of `load wait + rotate + tests` is the tile's arithmetic)")
print("   B detail: before sync", v[16] / (REP * B), "sync", v[17] / (REP * B), "loads+membership", v[18] / (REP * B), "chunks", v[19] / (REP * B))
print("   per scan: near run pairs", v[20] / max(v[23], 1), "runs", v[21] / max(v[23], 1), "ring entries", v[22] / max(v[23], 1))
mt = v[32:48].sum()
print("   merge_body's own stamps (share of D):", {k: round(v[32 + k] / max(mt, 1), 3) for k in range(1, 12) if v[32 + k]})
