"""the oracle is parity-unpinned (no PCL / Eigen / Boost in this image), so where it had to pick one reading of
an un-vendored dependency the other reading is a SWITCH (oracle/fx_oracle.h FXO_POLICY_*), and this test counts what each
switch moves on the golden fixtures' scans (tools/oracle_policies.py writes the full table: profiles/r05_oracle_policies.txt):

* the 3DSC zero-distance skip at FLT_EPSILON (SURVEY.md A.8-6's reading) instead of pcl::utils::equal's default tolerance
  numeric_limits<float>::min() (the oracle's and the kernels'): LIVE — on the 64- and 128-ring fixtures some keypoints have a
  cloud point within 0.35 mm, whose (huge, innermost-shell) weight the FLT_EPSILON reading drops.  The one reading a PCL run
  has to settle;
* Eigen 3.2's unguarded normalize() (a zero vector becomes NaN where 3.3 leaves it): inert — a NaN azimuth falls through the
  bin scan to the same fallback bin 0 that azimuth 0 selects;
* PCL >= 1.10's std::uniform_real_distribution<float>: inert — it differs from boost::uniform_01 narrowed to float only for
  draws >= 2^32 - 128, and the stream's first 6000 draws (2000 keypoints a scan) have none."""
import glob
import os

import numpy as np
import pytest

from tests import util
from tools.oracle_policies import moved

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
# rows (keypoints) whose descriptor the FLT_EPSILON skip changes, per fixture
EPSILON_ROWS = {"dense_128x2048_R2m_launch_seed10.npz": 16, "hdl64_64x2048_launch_seed10.npz": 38}


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_what_each_policy_moves_on_the_fixtures(oracle, path):
    g = np.load(path)
    name = os.path.basename(path)
    p, _lim, pts, roll, pitch = util.golden_case(g, name)
    base = oracle.run(p, pts, roll=roll, pitch=pitch)
    util.assert_bit_equal(base["descriptors"], g["descriptors"], f"{name}: policy 0 is the fixture")
    for pol, want in ((oracle.POLICY_SKIP_EPSILON, EPSILON_ROWS.get(name, 0)), (oracle.POLICY_EIGEN32_NORMALIZE, 0), (oracle.POLICY_STD_UNIFORM_FLOAT, 0)):
        r = oracle.run(p, pts, roll=roll, pitch=pitch, policy=pol)
        for k in ("keypoints", "kp_neighbors", "cand_keypoint"):  # (the policies live in the descriptor stage only)
            util.assert_bit_equal(r[k], base[k], f"{name} policy {pol:#x} {k}")
        values, rows, biggest = moved(r["descriptors"], base["descriptors"])
        assert rows == want and values == want, (name, hex(pol), values, rows, biggest)
        if want:  # one bin a row: the innermost radial shell's weight of the point the keypoint all but sits on
            assert biggest > 1.0


def test_the_epsilon_skip_drops_exactly_the_near_coincident_neighbours(oracle):
    """What the FLT_EPSILON reading changes, stated on the data: a row moves iff the keypoint has a cloud point at
    0 < d2 < FLT_EPSILON, and it loses that point's whole weight."""
    g = np.load([p for p in GOLDEN if "hdl64" in p][0])
    p, _lim, pts, roll, pitch = util.golden_case(g, "hdl64_64x2048_launch_seed10.npz")
    base = oracle.run(p, pts, roll=roll, pitch=pitch, want_rotated=True)
    eps = oracle.run(p, pts, roll=roll, pitch=pitch, policy=oracle.POLICY_SKIP_EPSILON)
    rot = base["rotated"][:, :3].astype(np.float32)
    near = []
    for k, kp in enumerate(base["keypoints"]):
        d = kp[None, :3] - rot
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        near.append(bool(((d2 > 0) & (d2 < np.finfo(np.float32).eps)).any()))
    changed = (eps["descriptors"] != base["descriptors"]).any(axis=1)
    assert changed.tolist() == near and sum(near) == 38
    assert (eps["descriptors"][changed] <= base["descriptors"][changed]).all()  # weights are only ever dropped


def test_the_rng_stream_has_no_draw_the_two_distributions_narrow_differently(oracle):
    u32, f32 = oracle.sc3d_rng(6000)
    assert int(u32.max()) < 2 ** 32 - 128  # (float)(u / 2^32) == 1.0f only from there: where std::uniform_real_distribution<float> differs
    assert np.array_equal(f32, (u32.astype(np.float64) / 2.0 ** 32).astype(np.float32))
