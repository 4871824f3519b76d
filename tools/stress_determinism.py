"""Race hunt: the same batch many times through one context; any bit difference between runs
(or against the oracle for the first scans) is reported with the stage it first shows up in."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_extraction_amd import capi
from oracle import oracle_py as O
from tests import util
preset = sys.argv[1] if len(sys.argv) > 1 else "launch"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
B = 64
scans = [capi.synth_scan(capi.synth_cfg(1000 + b)) for b in range(B)]
rng = np.random.default_rng(0)
bad = rng.choice(28800, 600, replace=False)
scans[3][bad[:300], 0] = np.nan
scans[5] = scans[5][rng.permutation(28800)]
p = capi.params(preset)
ctx = capi.Context(p, capi.limits(B, 28800))
ref = None
KEYS = ("filtered", "candidates", "cand_size", "cand_keypoint", "kpc", "kpc_cand", "keypoints", "kp_size", "kp_neighbors", "descriptors")
nbad = 0
for rep in range(reps):
    got = ctx.process_host(scans, roll=0.02, pitch=-0.015)
    if ref is None:
        ref = got
        for b in (0, 3, 5):
            try:
                util.compare_scan(got[b], O.run(p, scans[b], roll=0.02, pitch=-0.015), tag=f"scan {b}")
            except AssertionError as e:
                print("ORACLE MISMATCH", str(e)[:300]); nbad += 1
        continue
    for b in range(B):
        for k in KEYS:
            a, c = np.asarray(got[b][k]), np.asarray(ref[b][k])
            if a.shape != c.shape or not np.array_equal(a.view(np.uint32) if a.dtype.kind == "f" else a, c.view(np.uint32) if c.dtype.kind == "f" else c):
                print(f"rep {rep} scan {b}: {k} differs (shape {a.shape} vs {c.shape}) flags {got[b]['flags']:#x}/{ref[b]['flags']:#x}")
                nbad += 1
                break
print("mismatching (rep, scan) pairs:", nbad)
