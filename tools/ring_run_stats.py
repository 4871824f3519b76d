"""Diagnostic: runs and segments per ring of the headline workload (what the run tier's table sizes have to hold)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from feature_extraction_amd import capi
B = 64
p = capi.params("launch")
scans = [capi.synth_scan(capi.synth_cfg(1000 + b)) for b in range(B)]
ctx = capi.Context(p, capi.limits(B, 28800))
got = ctx.process_host(scans, roll=0.02, pitch=-0.015, debug=True)
tol2 = np.float32(float(p.cluster_tolerance) ** 2)
runs, npts = [], []
for g in got:
    f = g["filtered"]
    el = f[:, 3]
    for r in range(16):
        c = p.el0_deg + r * p.el_step_deg
        m = (el >= np.float32(c - p.el_step_deg / 2)) & (el <= np.float32(c + p.el_step_deg / 2))
        q = f[m, :3]
        if len(q) == 0:
            continue
        d2 = ((q[1:] - q[:-1]) ** 2).sum(1)
        runs.append(1 + int((~(d2 < tol2)).sum()))
        npts.append(len(q))
runs, npts = np.array(runs), np.array(npts)
for S in (64, 96, 128):
    seg_len = np.maximum(8, (npts + (S * 3 // 4) - 1) // (S * 3 // 4))
    segs_ub = npts // seg_len + runs
    print(f"S = RN = {S}: rings with more runs than RN {np.mean(runs > S):.3%}, with (points / seg_len + runs) > S {np.mean(segs_ub > S):.3%}")
print("points per ring: median", np.median(npts), "p90", np.percentile(npts, 90), "max", npts.max())
print("runs per ring: median", np.median(runs), "p90", np.percentile(runs, 90), "p99", np.percentile(runs, 99), "max", runs.max())
