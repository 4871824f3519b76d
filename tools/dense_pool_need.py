"""Diagnostic: how many dense-tier pool entries the VLP-16 fuzz scenes need (sizing of the default max_dense_points)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from feature_extraction_amd import capi
from tests.test_gpu_fuzz import _case
lib = capi.load()
lib.fx_debug_counters.argtypes = [C.c_void_p, C.c_void_p]
need = []
for seed in list(range(20000, 22000)) + list(range(7000, 7800)):
    s, p, roll, pitch, what = _case(seed)
    ctx = capi.Context(p, capi.limits(1, 28800, max_candidates=3500, max_kpc_points=57600, max_keypoints=1024, max_total_keypoints=1024, max_ring_candidates=512, max_dense_points=30_000_000))
    got = ctx.process_host([s], roll=roll, pitch=pitch)[0]
    cnt = (C.c_uint32 * 16)()
    capi.check(lib.fx_debug_counters(ctx.handle, cnt))
    need.append((int(cnt[13]), seed, got["flags"], got["n_keypoints"], what["over"]["descriptor_radius"]))
    ctx.close()
need.sort(reverse=True)
print(need[:12])
n = np.array([x[0] for x in need])
for m in (1, 4, 16, 32, 64, 128, 256):
    print(f"pool of {m} scans' worth ({m * 28800}): {int((n > m * 28800).sum())} of {len(n)} cases exceed it")
