"""The one LIVE reading of PCL the oracle had to choose: pcl::ShapeContext3DEstimation skips a
neighbour that coincides with the keypoint — at d^2 < numeric_limits<float>::min() (pcl::utils::equal's default tolerance:
what the oracle and the product follow) or at d^2 < FLT_EPSILON (SURVEY.md A.8-6's reading).  Only a run of real PCL settles
it (tools/pcl_crosscheck); until then BOTH sides carry the other reading as one switch — the oracle's FXO_POLICY_SKIP_EPSILON,
the device's -DFX_SKIP_EPSILON (lib/libfx_hip_skipeps.so: the constant in sc3d_is_origin) — and this test proves they flip
TOGETHER: on the fixtures where the reading moves values (the 64- and 128-ring ones: a cloud point within 0.35 mm of a keypoint)
the measurement build equals the oracle under the switch, differs from the default exactly where the oracle does, and on the
VLP-16 fixtures nothing moves on either side."""
import glob
import os

import numpy as np
import pytest

from feature_extraction_amd import build, capi
from tests import util

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))


def _run(lib, p, lim, pts, roll, pitch):
    saved = capi.LIB_PATH, capi._lib
    capi.LIB_PATH, capi._lib = lib, None
    try:
        ctx = capi.Context(p, lim)
        got = ctx.process_host([pts], roll=roll, pitch=pitch)[0]
        ctx.close()
    finally:
        capi.LIB_PATH, capi._lib = saved
    return got


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_device_and_oracle_flip_together(fxlib, oracle, path):
    name = os.path.basename(path)[:-4]
    p, lim, pts, roll, pitch = util.golden_case(np.load(path), name)
    o_min = oracle.run(p, pts, roll=roll, pitch=pitch)
    o_eps = oracle.run(p, pts, roll=roll, pitch=pitch, policy=oracle.POLICY_SKIP_EPSILON)
    g_min = _run(capi.LIB_PATH, p, lim, pts, roll, pitch)
    g_eps = _run(build.build_skip_epsilon(), p, lim, pts, roll, pitch)
    util.compare_scan(g_min, o_min, tag=f"{name}: default reading")
    util.compare_scan(g_eps, o_eps, tag=f"{name}: FLT_EPSILON reading")  # detector integer-exact, descriptors within DESC_TOL
    d_o = np.nan_to_num(o_min["descriptors"]) != np.nan_to_num(o_eps["descriptors"])
    d_g = np.nan_to_num(g_min["descriptors"]) != np.nan_to_num(g_eps["descriptors"])
    assert np.array_equal(d_o, d_g), f"{name}: the device and the oracle move different values under the FLT_EPSILON reading"
    rows = int(d_o.any(axis=1).sum())
    print(f"{name}: the FLT_EPSILON reading moves {int(d_o.sum())} values in {rows} of {len(d_o)} rows — on the device and in the oracle alike")
    if name.startswith(("hdl64", "dense_128")):
        assert rows > 0  # (profiles/r05_oracle_policies.txt: 38 of 158 / 16 of 158 rows)
    else:
        assert rows == 0
