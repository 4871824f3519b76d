"""The phi / theta policy (fp64 rounded once, in the oracle and in the product) against PCL's literal fp32
atan2f / acosf, on the five golden fixtures: the measurement build of the product (lib/libfx_hip_trigf32.so,
-DFX_TRIG_LITERAL_F32: the device's atan2f / acosf, no exact re-evaluation next to a bin edge) must give the default build's
descriptors except for whole weights that hop between ADJACENT angular bins — and the test prints every such hop.  The oracle
gets the same treatment with glibc's atan2f / acosf.  (tools/trig_policy.py counts the same on more scans: profiles/r04_trig_policy.txt.)"""
import glob
import os

import numpy as np
import pytest

from feature_extraction_amd import build, capi
from tests import util

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))


def _hops(a, b, tag):
    """Values that differ between the two policies.  Each must be a weight leaving one bin for a neighbour in azimuth (+-1 of
    12, cyclic) or elevation (+-1 of 11) at the same radius bin: the row's total must not change."""
    a, b = np.nan_to_num(a[:, :capi.FX_DESC_BINS]), np.nan_to_num(b[:, :capi.FX_DESC_BINS])
    moved = np.argwhere(np.abs(a - b) > util.DESC_TOL)
    for r in sorted(set(moved[:, 0])):
        bins = moved[moved[:, 0] == r, 1]
        print(f"{tag}: descriptor {r}: bins {[(int(x), float(a[r, x]), float(b[r, x])) for x in bins]}")
        assert abs(float(a[r].sum()) - float(b[r].sum())) <= 1e-3 * max(1.0, float(a[r].sum())), f"{tag}: descriptor {r} lost or gained weight"
        for x in bins:
            l, k, j = x // 165, x % 165 // 15, x % 15
            partners = [y for y in bins if y != x and y % 15 == j and ((y // 165 - l) % 12 in (0, 1, 11)) and abs(y % 165 // 15 - k) <= 1]
            assert partners, f"{tag}: descriptor {r} bin {x} changed without an adjacent partner"
    return len(moved)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_literal_fp32_angles_move_at_most_whole_weights_between_adjacent_bins(fxlib, oracle, path):
    name = os.path.basename(path)[:-4]
    p, lim, pts, roll, pitch = util.golden_case(np.load(path), name)
    # the oracle under both policies (glibc's atan2f / acosf)
    o64 = oracle.run(p, pts, roll=roll, pitch=pitch)
    o32 = oracle.run(p, pts, roll=roll, pitch=pitch, trig=oracle.TRIG_LIBM_F32)
    assert o64["n_keypoints"] == o32["n_keypoints"]
    n_o = _hops(o64["descriptors"], o32["descriptors"], f"{name} oracle")
    # the product under both policies
    out = []
    for lib in (capi.LIB_PATH, build.build_trig_literal()):
        saved = capi.LIB_PATH, capi._lib
        capi.LIB_PATH, capi._lib = lib, None
        try:
            ctx = capi.Context(p, lim)
            got = ctx.process_host([pts], roll=roll, pitch=pitch)[0]
            ctx.close()
        finally:
            capi.LIB_PATH, capi._lib = saved
        assert got["flags"] == 0 and got["n_keypoints"] == o64["n_keypoints"]
        util.assert_bit_equal(got["keypoints"], o64["keypoints"], f"{name} keypoints")
        out.append(got["descriptors"])
    n_p = _hops(out[0], out[1], f"{name} product")
    print(f"{name}: {n_p} values move in the product (device atan2f / acosf), {n_o} in the oracle (glibc), of "
          f"{int((np.nan_to_num(out[0]) != 0).sum())} non-empty")
    # and the default build is the oracle's policy
    assert np.abs(np.nan_to_num(out[0]) - np.nan_to_num(o64["descriptors"])).max(initial=0.0) <= util.DESC_TOL
