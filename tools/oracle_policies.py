"""How many descriptor values each switchable reading of the un-vendored dependencies would move (oracle/fx_oracle.h
FXO_POLICY_*: the 3DSC zero-distance skip at FLT_EPSILON instead of numeric_limits<float>::min(); Eigen 3.2's unguarded
normalize(); PCL >= 1.10's std::uniform_real_distribution<float> x-axis draws) — on the five golden fixtures' scans and on
256-pole VLP-16 scenes.  CPU only (the oracle against itself).  Whoever runs tools/pcl_crosscheck against a given PCL knows from
this table which switch can matter at all.
  python tools/oracle_policies.py [out.txt]"""
import glob
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_extraction_amd import capi  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
from tests import util  # noqa: E402

POLICIES = (("skip d2 < FLT_EPSILON", O.POLICY_SKIP_EPSILON), ("Eigen 3.2 normalize()", O.POLICY_EIGEN32_NORMALIZE),
            ("std::uniform_real_distribution<float>", O.POLICY_STD_UNIFORM_FLOAT),
            ("all three", O.POLICY_SKIP_EPSILON | O.POLICY_EIGEN32_NORMALIZE | O.POLICY_STD_UNIFORM_FLOAT))


def moved(a, b):
    """(values that differ, rows that differ, max |diff|) between two descriptor arrays (NaN == NaN)."""
    both_nan = np.isnan(a) & np.isnan(b)
    diff = (a != b) & ~both_nan
    mag = np.abs(np.where(diff, np.nan_to_num(a) - np.nan_to_num(b), 0.0))
    return int(diff.sum()), int(diff.any(axis=1).sum()) if len(a) else 0, float(mag.max(initial=0.0))


def cases():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for path in sorted(glob.glob(os.path.join(here, "tests", "golden", "*.npz"))):
        g = np.load(path)
        p, _lim, pts, roll, pitch = util.golden_case(g, os.path.basename(path))
        yield os.path.basename(path), p, pts, roll, pitch
    for seed in (1, 2, 3, 4):
        yield f"vlp16 256 poles seed {seed}", capi.params("launch"), capi.synth_scan(capi.synth_cfg(seed, n_poles=256)), 0.02, -0.015


def main(out_path=None):
    lines = ["case, keypoints, descriptor values | per policy: values moved / rows moved / max |diff|"]
    for name, p, pts, roll, pitch in cases():
        base = O.run(p, pts, roll=roll, pitch=pitch)
        row = [name, str(base["n_keypoints"]), str(base["descriptors"].size)]
        for label, pol in POLICIES:
            r = O.run(p, pts, roll=roll, pitch=pitch, policy=pol)
            assert r["n_keypoints"] == base["n_keypoints"]
            v, rows, mx = moved(r["descriptors"], base["descriptors"])
            row.append(f"{label}: {v} / {rows} / {mx:.3g}")
        lines.append(" | ".join(row))
        print(lines[-1], flush=True)
    if out_path:
        open(out_path, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
