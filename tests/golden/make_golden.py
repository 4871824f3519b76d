#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz: inputs + the oracle's outputs for them.

The reference ships no test data and PCL cannot run here, so these fixtures
are produced by the CPU oracle (oracle/fx_oracle.cpp, brute-force search) on scans from the
product's synthetic generator.  They pin (a) the oracle against accidental change and
(b) the HIP path against the oracle on the GPU box, where /root/reference does not exist.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from feature_extraction_amd import capi  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {
    # name: (preset, seed, roll, pitch, synth overrides)
    "vlp16_default_seed1000": ("default", 1000, 0.02, -0.015, {}),
    "vlp16_launch_seed1000": ("launch", 1000, 0.02, -0.015, {}),
    "vlp16_launch_seed1001_unleveled": ("launch", 1001, 0.0, 0.0, {}),
}
# BASELINE.json configs 3 and 5, one scan each (the first scene bench.py's other_configs cycle).  Their inputs are not
# stored (2 and 4 MB): the fixture carries the generator's configuration and the test regenerates the scan; the
# outputs are stored.  "spec" = everything a reader needs to rebuild parameters, limits and input.
import json  # noqa: E402
BIG_CASES = {
    "hdl64_64x2048_launch_seed10": dict(
        preset="launch", seed=10, roll=0.02, pitch=-0.015,
        synth=dict(n_rings=64, n_az=2048, el0_deg=-24.8, el_step_deg=26.8 / 63, n_poles=256),
        params=dict(n_rings=64, el0_deg=-24.8, el_step_deg=26.8 / 63, secondary_max=64),
        limits=dict(max_candidates=4096, max_kpc_points=32768, max_keypoints=512, max_total_keypoints=512)),
    "dense_128x2048_R2m_launch_seed10": dict(
        preset="launch", seed=10, roll=0.02, pitch=-0.015,
        synth=dict(n_rings=128, n_az=2048, el0_deg=-25.0, el_step_deg=40.0 / 127, n_poles=256),
        params=dict(n_rings=128, el0_deg=-25.0, el_step_deg=40.0 / 127, secondary_max=128, descriptor_radius=2.0),
        limits=dict(max_candidates=8192, max_kpc_points=65536, max_keypoints=512, max_total_keypoints=512)),
}
KEYS = ("filtered", "candidates", "cand_size", "cand_keypoint", "kpc", "kpc_cand", "keypoints", "kp_size",
        "kp_neighbors", "descriptors")


def main():
    for name, (preset, seed, roll, pitch, over) in CASES.items():
        pts = capi.synth_scan(capi.synth_cfg(seed, **over))
        p = capi.params(preset)
        r = O.run(p, pts, roll=roll, pitch=pitch, search=O.SEARCH_BRUTE)
        out = {k: r[k] for k in KEYS}
        out["points_xyz"] = pts[:, :3].copy()
        out["meta"] = np.array([seed, roll, pitch], np.float64)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: N={len(pts)} N_f={len(r['filtered'])} C={len(r['candidates'])} K={r['n_keypoints']} "
              f"-> {os.path.getsize(path) / 1024:.0f} KiB")
    for name, c in BIG_CASES.items():
        pts = capi.synth_scan(capi.synth_cfg(c["seed"], **c["synth"]))
        p = capi.params(c["preset"], **c["params"])
        r = O.run(p, pts, roll=c["roll"], pitch=c["pitch"], search=O.SEARCH_KDTREE)
        out = {k: r[k] for k in KEYS}
        out["spec"] = np.frombuffer(json.dumps(c).encode(), dtype=np.uint8)
        out["points_checksum"] = np.array([int(pts.view(np.uint32).astype(np.uint64).sum())], np.uint64)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: N={len(pts)} N_f={len(r['filtered'])} C={len(r['candidates'])} K={r['n_keypoints']} "
              f"max neighbours {int(r['kp_neighbors'].max())} -> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
