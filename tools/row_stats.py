"""Diagnostic: how full the descriptor rows of the headline workload are (NaN rows, non-zero bins per row)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from feature_extraction_amd import capi
B = 64
scans = [capi.synth_scan(capi.synth_cfg(1000 + b)) for b in range(B)]
ctx = capi.Context(capi.params("launch"), capi.limits(B, 28800))
got = ctx.process_host(scans, roll=0.02, pitch=-0.015)
rows = nan = 0
nb = []
for g in got:
    d = g["descriptors"]
    rows += len(d)
    nan += int(np.isnan(d[:, 0]).sum())
    nb += list((d[:, :1980] != 0).sum(1)[~np.isnan(d[:, 0])])
nb = np.array(nb)
print("rows", rows, "NaN rows", nan, "non-zero bins per row: median", np.median(nb), "mean", nb.mean(), "p90", np.percentile(nb, 90), "max", nb.max())
