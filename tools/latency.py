"""Streaming latency: one fx_process_batch call per scan (or per small batch),
host buffer in -> keypoints + descriptors back in pinned host memory, as the ROS shell would call it.

  python tools/latency.py [batch=1] [calls=300] [preset=launch] [graph=0|1] [stride_bytes=16|32]
(stride 32: pcl::PointXYZI records in host memory, what the reference-side binding has — ref: node.cpp:81)
"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_extraction_amd import capi

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 300
preset = sys.argv[3] if len(sys.argv) > 3 else "launch"
graph = int(sys.argv[4]) if len(sys.argv) > 4 else 0
stride = int(sys.argv[5]) if len(sys.argv) > 5 else 16
if os.environ.get("FX_LIB"):  # (an alternative build under feature_extraction_amd/lib: A/B runs)
    capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), os.environ["FX_LIB"])
capi.load()
scans = [capi.synth_scan(capi.synth_cfg(1000 + b)) for b in range(max(B, 16))]
if stride == 32:
    wide = []
    for s in scans:
        w = np.zeros((len(s), 8), np.float32)
        w[:, :3] = s[:, :3]
        wide.append(w)
    scans = wide
ctx = capi.Context(capi.params(preset), capi.limits(B, 28800))
ctx.set_graph_batch(B if graph else 0)
sets = []
for j in range(len(scans) // B):
    part = scans[j * B:(j + 1) * B]
    sets.append(ctx.make_descs([s.ctypes.data for s in part], [len(s) for s in part], stride, 0.02, -0.015))
modes = [("host->host (keypoints + descriptors)", capi.FX_OUT_HOST)]
if os.environ.get("FX_LATENCY_SPLIT"):  # (what the copies back cost: the same call with its results left on the device, then a stream sync)
    modes.append(("host->device (results stay in HBM) + sync", 0))
for mode, flags in modes:
    for w in range(20):
        ctx.process_raw(sets[w % len(sets)], B, flags)
    ctx.synchronize()
    t = np.empty(calls)
    for i in range(calls):
        t0 = time.perf_counter()
        v = ctx.process_raw(sets[i % len(sets)], B, flags)
        if not flags:
            ctx.synchronize()
        t[i] = time.perf_counter() - t0
    t *= 1e3
    print(json.dumps({"mode": mode, "batch": B, "calls": calls, "preset": preset, "graph": graph, "stride_bytes": stride, "ms_median": float(np.median(t)),
                      "ms_p99": float(np.percentile(t, 99)), "ms_min": float(t.min()),
                      "keypoints_last": int(v.total_keypoints)}))
# where the time goes: device spans of the stages (HIP events) for the same call pattern
ctx.set_graph_batch(0)
ctx.set_profiling(1)
acc = {n: 0.0 for n in capi.STAGE_NAMES}; tot = 0.0
for i in range(50):
    ctx.process_raw(sets[i % len(sets)], B, capi.FX_OUT_HOST)
    ms, total = ctx.timings(0)
    for n in ms:
        acc[n] += ms[n]
    tot += total
print(json.dumps({"stage_ms": {n: round(a / 50, 4) for n, a in acc.items()}, "stages_total_ms": round(tot / 50, 4)}))
