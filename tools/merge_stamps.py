"""Diagnostic: per-phase cycles of the merge tier (merge_body: k_merge_small / big / huge) on one of BASELINE's configurations
(needs lib/libfx_hip_stamps.so built with -DFX_STAMPS).  Usage on the GPU box: python tools/merge_stamps.py [2|3|5]"""
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

which = sys.argv[1] if len(sys.argv) > 1 else "5"
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
base = 32
names = {1: "candidates in, bin counts", 2: "cell sort", 3: "pair tests + unions", 4: "roots, sizes", 5: "cc_order a", 6: "cc_order b", 7: "cc_order c",
         8: "member lists, centroids", 9: "candidate -> keypoint", 10: "keypoint_cloud bases", 11: "keypoint_cloud copy"}
tot = sum(v[base + k] for k in names)
print(f"{name}: {B} scans; cycles (100 MHz clock) summed over the workgroups' lane 0")
for k, nm in names.items():
    print(f"   {nm:28s} {v[base + k] / max(tot, 1) * 100:6.2f} %   {v[base + k] / B:10.0f} per scan")
if v[60]:
    print(f"k_merge_huge pair loop: {v[60] / B:.0f} wavefront iterations (a source against 64 targets) per scan, {v[61] / B:.0f} pairs within the tolerance, "
          f"{v[62] / B:.0f} of them not recognised as one set already (union calls)")
