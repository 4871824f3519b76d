#!/usr/bin/env python3
"""Writes the golden fixtures (tests/golden/*.npz) in a form a plain C++ program can read:
  out/NAME.pcd            binary PCD, fields x y z intensity (float32), the fixture's input scan
  out/NAME.txt            key value lines: the 14 node parameters, n_rings / el0_deg / el_step_deg / secondary_max, roll, pitch
  out/NAME_expected.bin   u32 counts {N_f, C, K}, then filtered[N_f][4], keypoints_full[C][4], keypoints[K][4],
                          kp_neighbors[K] (u32), descriptors[K][1989] — all float32 unless noted, the oracle's outputs
usage: python tools/pcl_crosscheck/export_fixtures.py [out_dir]"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import util  # noqa: E402

out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "out")
os.makedirs(out_dir, exist_ok=True)
for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz"))):
    name = os.path.basename(path)[:-4]
    g = np.load(path)
    p, _lim, pts, roll, pitch = util.golden_case(g, name + ".npz")
    n = len(pts)
    with open(os.path.join(out_dir, name + ".pcd"), "wb") as f:
        f.write((f"# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\n"
                 f"COUNT 1 1 1 1\nWIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA binary\n").encode())
        f.write(np.ascontiguousarray(pts, np.float32).tobytes())
    with open(os.path.join(out_dir, name + ".txt"), "w") as f:
        for k in ("x_min", "x_max", "y_min", "y_max", "z_min", "z_max", "cluster_tolerance", "cluster_min_count", "cluster_max_count",
                  "cluster_radius_threshold", "number_detection_channels", "estimate_descriptors", "descriptor_radius", "n_rings", "el0_deg",
                  "el_step_deg", "secondary_max"):
            f.write(f"{k} {getattr(p, k)!r}\n")
        f.write(f"roll {float(roll)!r}\npitch {float(pitch)!r}\n")
    with open(os.path.join(out_dir, name + "_expected.bin"), "wb") as f:
        f.write(np.array([len(g["filtered"]), len(g["candidates"]), len(g["keypoints"])], np.uint32).tobytes())
        for key, dt in (("filtered", np.float32), ("candidates", np.float32), ("keypoints", np.float32), ("kp_neighbors", np.uint32),
                        ("descriptors", np.float32)):
            f.write(np.ascontiguousarray(g[key], dt).tobytes())
    print(f"{name}: {n} points, K = {len(g['keypoints'])}")
