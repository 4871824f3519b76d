"""Race hunt on the many-ring configurations: the same batch of dense scans many times through one context (and a second
context in flight beside it); any bit difference between runs is reported.  python tools/stress_dense.py [3|5] [reps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_extraction_amd import capi
import bench
which = sys.argv[1] if len(sys.argv) > 1 else "5"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
name = [n for n in bench.OTHER_CONFIGS if n.startswith(f"config{which}")][0]
cfg = bench.OTHER_CONFIGS[name]
B = 8
scans = [capi.synth_scan(capi.synth_cfg(10 + b, **cfg["synth"])) for b in range(B)]
p = capi.params(cfg["preset"], **cfg["params"])
lim = capi.limits(B, len(scans[0]), **dict(cfg["limits"], max_total_keypoints=B * 256))
ctx, other = capi.Context(p, lim), capi.Context(p, lim)
KEYS = ("filtered", "candidates", "cand_size", "cand_keypoint", "kpc", "kpc_cand", "keypoints", "kp_size", "kp_neighbors", "descriptors")
ref, nbad = None, 0
for rep in range(reps):
    other.process_host(scans[::-1], roll=0.02, pitch=-0.015) if rep % 2 else None  # (another context's batch now and then)
    got = ctx.process_host(scans if rep % 3 else scans[:5], roll=0.02, pitch=-0.015)  # (batch sizes change: rows keep their records)
    if ref is None:
        ref = ctx.process_host(scans, roll=0.02, pitch=-0.015)
    for b in range(len(got)):
        for k in KEYS:
            a, c = np.asarray(got[b][k]), np.asarray(ref[b][k])
            if a.shape != c.shape or not np.array_equal(a.view(np.uint32) if a.dtype.kind == "f" else a, c.view(np.uint32) if c.dtype.kind == "f" else c):
                print(f"rep {rep} scan {b}: {k} differs (shape {a.shape} vs {c.shape}) flags {got[b]['flags']:#x}/{ref[b]['flags']:#x}")
                nbad += 1
                break
print(f"{name}: {reps} repetitions of {B} scans, keypoints {sum(len(r['keypoints']) for r in ref)}, mismatching (rep, scan) pairs: {nbad}")
