"""Diagnostic: is the throughput mode bound by the host thread that enqueues the batches?  Times the enqueue loop of K
batches over C contexts (before any synchronisation) against the time to completion.
  python tools/host_enqueue.py [contexts] [steps] [batch]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_extraction_amd import capi

C_ = int(sys.argv[1]) if len(sys.argv) > 1 else 3
K = int(sys.argv[2]) if len(sys.argv) > 2 else 60
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
capi.load()
uniq = [capi.synth_scan(capi.synth_cfg(1000 + b)) for b in range(64)]
dev = [torch.from_numpy(s).cuda() for s in uniq]
N = len(uniq[0])
ctxs = [capi.Context(capi.params("launch"), capi.limits(B, N)) for _ in range(C_)]
descs = ctxs[0].make_descs([dev[b % 64].data_ptr() for b in range(B)], [N] * B, 16, 0.02, -0.015)
for _ in range(3):
    for c in ctxs:
        c.process_raw(descs, B, capi.FX_IN_DEVICE)
for c in ctxs:
    c.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(K):
        ctxs[i % C_].process_raw(descs, B, capi.FX_IN_DEVICE)
    t1 = time.perf_counter()
    for c in ctxs:
        c.synchronize()
    t2 = time.perf_counter()
    print(f"contexts {C_} batch {B}: enqueue {1e3 * (t1 - t0) / K:.3f} ms/batch, complete {1e3 * (t2 - t0) / K:.3f} ms/batch "
          f"-> {B * K / (t2 - t0):.0f} scans/s")
