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


if __name__ == "__main__":
    main()
