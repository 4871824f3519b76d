"""Diagnostic: is the headline bound by the ONE host thread that enqueues the batches?  T host threads, each with its own
context(s) on its own stream(s), enqueue batches as fast as they can for a fixed number of steps (ctypes releases the GIL
during fx_process_batch).
  python tools/host_threads.py [threads] [contexts per thread] [steps per context] [batch]"""
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_extraction_amd import capi  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
CPT = int(sys.argv[2]) if len(sys.argv) > 2 else 1
K = int(sys.argv[3]) if len(sys.argv) > 3 else 100
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
capi.load()
uniq = [capi.synth_scan(capi.synth_cfg(1000 + b)) for b in range(128)]
dev = [torch.from_numpy(s).cuda() for s in uniq]
N = len(uniq[0])
ctxs = [capi.Context(capi.params("launch"), capi.limits(B, N, sparse=True)) for _ in range(T * CPT)]
descs = [ctxs[0].make_descs([dev[(b + o) % 128].data_ptr() for b in range(B)], [N] * B, 16, 0.02, -0.015) for o in (0, 64)]
for _ in range(3):
    for c in ctxs:
        c.process_raw(descs[0], B, capi.FX_IN_DEVICE)
for c in ctxs:
    c.synchronize()


def worker(mine, enq):
    t0 = time.perf_counter()
    for i in range(K * len(mine)):
        mine[i % len(mine)].process_raw(descs[(i // len(mine)) % 2], B, capi.FX_IN_DEVICE)
    enq.append(time.perf_counter() - t0)
    for c in mine:
        c.synchronize()


for rep in range(3):
    enq = []
    th = [threading.Thread(target=worker, args=(ctxs[t * CPT:(t + 1) * CPT], enq)) for t in range(T)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    n = K * T * CPT
    print(f"threads {T} x contexts {CPT}, batch {B}: {1e3 * dt / n:.4f} ms/batch -> {B * n / dt:.0f} scans/s; a thread's enqueue loop {1e3 * max(enq) / (K * CPT):.4f} ms/batch", flush=True)
