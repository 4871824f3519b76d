"""Experiment: throughput with the batch replayed as one HIP graph (no per-stage events)."""
import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
import numpy as np, torch
from feature_extraction_amd import capi
import bench
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
B, N = 1024, 28800
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
use_graph = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
scans = bench.make_scans(capi, [1000 + b for b in range(B)], 64)
d_in = torch.from_numpy(np.stack(scans)).to(dev)
scans_b = bench.make_scans(capi, [1000 + B + b for b in range(B)], 64)
d_in_b = torch.from_numpy(np.stack(scans_b)).to(dev)
p = capi.params("launch")
ctxs = [capi.Context(p, capi.limits(B, N)) for _ in range(K)]
for c in ctxs:
    c.set_graph_batch(B if use_graph else 0)
descs = ctxs[0].make_descs([d_in.data_ptr() + b * N * 16 for b in range(B)], [N] * B, 16, 0.02, -0.015)
descs_b = ctxs[0].make_descs([d_in_b.data_ptr() + b * N * 16 for b in range(B)], [N] * B, 16, 0.02, -0.015)
cnt = [0]
def step():
    j = cnt[0] % K
    d = descs_b if (cnt[0] // K) % 2 else descs
    cnt[0] += 1
    ctxs[j].process_raw(d, B, capi.FX_IN_DEVICE)
for _ in range(5 * K): step()
torch.cuda.synchronize()
res = []
for rep in range(15):
    t0 = time.perf_counter()
    for _ in range(100): step()
    for c in ctxs: c.synchronize()
    res.append(time.perf_counter() - t0)
print(f"contexts {K} graph {use_graph}: median {B * 100 / np.median(res):.0f} scans/s")
