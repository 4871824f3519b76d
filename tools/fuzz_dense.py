"""Diagnostic: differential fuzz on dense many-ring sensors (32 / 64 rings, 512-1024 azimuths): random scenes x random node
parameters against the oracle.  These scans exercise what the VLP-16 fuzz does not: the second run tier, the workgroup ring
tier, the large merge tier, the long-list, whole-CU and slab descriptor tiers.
  python tools/fuzz_dense.py FIRST LAST"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_extraction_amd import capi
from oracle import oracle_py as O
from tests import util

capi.load()
O.load()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
import ctypes as C
bad = total_k = flagged = 0
tiers = np.zeros(16, np.int64)
for seed in range(lo, hi):
    rng = np.random.default_rng(seed)
    R = int(rng.choice([32, 64]))
    n_az = int(rng.choice([512, 768, 1024]))
    el_span = float(rng.uniform(20.0, 40.0))
    el0 = -float(rng.uniform(12.0, 25.0))
    cfg = capi.synth_cfg(int(rng.integers(1, 1 << 30)), n_rings=R, n_az=n_az, el0_deg=el0, el_step_deg=el_span / (R - 1),
                         n_poles=int(rng.integers(0, 200)), pole_radius=float(rng.uniform(0.03, 0.4)),
                         sensor_height=float(rng.uniform(1.0, 2.5)), wall_radius=float(rng.uniform(30, 100)))
    s = capi.synth_scan(cfg)
    over = dict(n_rings=R, el0_deg=el0, el_step_deg=el_span / (R - 1), secondary_max=R,
                cluster_tolerance=float(rng.uniform(0.15, 1.2)), cluster_min_count=int(rng.integers(1, 6)),
                cluster_max_count=int(rng.integers(20, 1500)), cluster_radius_threshold=float(rng.uniform(0.08, 0.5)),
                number_detection_channels=int(rng.integers(1, 6)), descriptor_radius=float(rng.uniform(0.5, 3.0)),
                cloud_leveling=int(rng.integers(0, 2)))
    p = capi.params(str(rng.choice(["default", "launch"])), **over)
    if rng.integers(0, 4) == 0:  # azimuth blocks out of order
        blocks = np.array_split(np.arange(len(s)), 24)
        s = np.ascontiguousarray(s[np.concatenate([blocks[i] for i in rng.permutation(len(blocks))])])
    roll, pitch = float(rng.uniform(-0.05, 0.05)), float(rng.uniform(-0.05, 0.05))
    ctx = capi.Context(p, capi.limits(1, len(s), max_candidates=8192, max_kpc_points=2 * len(s), max_keypoints=1024, max_total_keypoints=1024,
                                      max_ring_candidates=1024))
    got = ctx.process_host([s], roll=roll, pitch=pitch)[0]
    cnt = (C.c_uint32 * 16)()
    ctx.lib.fx_debug_counters.argtypes = [C.c_void_p, C.c_void_p]
    capi.check(ctx.lib.fx_debug_counters(ctx.handle, cnt))
    tiers += (np.array(list(cnt)) > 0)
    ctx.close()
    if got["flags"]:
        flagged += 1
        print("flagged", seed, hex(got["flags"]), R, n_az)
        continue
    ora = O.run(p, s, roll=roll, pitch=pitch)
    try:
        st = util.compare_scan(got, ora, tag=f"seed {seed}")
        total_k += st["K"]
    except AssertionError as e:
        bad += 1
        print("MISMATCH", seed, str(e)[:300], dict(R=R, n_az=n_az, over=over))
print(f"dense seeds {lo}..{hi}: {bad} mismatches, {flagged} flagged, {total_k} keypoints, {time.time() - t0:.0f} s")
names = ["second run tier", "big merge", "re-gather", "-", "list rows", "workgroup ring tier", "whole-CU rows", "exact-angle rows", "wave rows", "huge merge",
         "-", "-", "slab rows"]
print("cases that used: " + ", ".join(f"{n} {int(c)}" for n, c in zip(names, tiers) if n != "-"))
