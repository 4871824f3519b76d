"""Experiment: S contexts on S streams, each with batch/S scans — do the latency-bound stage
kernels of different sub-batches overlap?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_extraction_amd import capi
B, N = 1024, 28800
scans = np.stack([capi.synth_scan(capi.synth_cfg(1000 + b)) for b in range(128)])
host = np.concatenate([scans] * (B // 128))
d = torch.from_numpy(host).cuda()
p = capi.params("launch")
for S in (1, 2, 4, 8):
    ctxs, descs, streams = [], [], []
    sub = B // S
    for s in range(S):
        c = capi.Context(p, capi.limits(sub, N))
        st = torch.cuda.Stream()
        c.set_stream(st.cuda_stream)
        ctxs.append(c); streams.append(st)
        descs.append(c.make_descs([d.data_ptr() + (s * sub + b) * N * 16 for b in range(sub)], [N] * sub, 16, 0.02, -0.015))
    def step():
        for c, ds in zip(ctxs, descs):
            c.process_raw(ds, sub, capi.FX_IN_DEVICE)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K): step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"streams {S}: {B * K / dt:,.0f} scans/s  ({dt / K * 1e3:.3f} ms per {B} scans)")
    for c in ctxs: c.close()
