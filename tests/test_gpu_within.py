"""The descriptor tiers count 3DSC's local point density two support points per packed instruction, the comparison folded
into `v_pk_fma_f32 ... clamp` of the exactly scaled difference (WithinR2 in csrc/fx_kernels.hip).  It must give what the plain
`dist2(...) < r2` gives — also when a distance equals the radius to the bit, is one ulp either side of it, overflows, or the
radius is tiny or huge — and that is what numpy gives in FLANN's operation order."""
import numpy as np
import pytest

from feature_extraction_amd import capi

pytestmark = pytest.mark.gpu


def _device(sup, qry, r2):
    lib = capi.load_test()  # (the fx_test_* entry points exist only in the test build)
    sup = np.ascontiguousarray(sup, np.float32)
    qry = np.ascontiguousarray(qry, np.float32)
    a, c = np.empty(len(qry), np.uint32), np.empty(len(qry), np.uint32)
    capi.check(lib.fx_test_within_device(0, sup.ctypes.data, len(sup), qry.ctypes.data, len(qry), np.float32(r2), a.ctypes.data, c.ctypes.data))
    return a, c


def _numpy(sup, qry, r2):
    out = np.empty(len(qry), np.uint32)
    for i, b in enumerate(qry):
        d = b[None, :3].astype(np.float32) - sup[:, :3].astype(np.float32)
        r = d[:, 0] * d[:, 0]
        r = r + d[:, 1] * d[:, 1]
        r = r + d[:, 2] * d[:, 2]
        out[i] = np.count_nonzero(r < np.float32(r2))
    return out


@pytest.mark.parametrize("scale", [1.0, 1e-6, 3e4])
def test_packed_count_equals_the_plain_compare(scale):
    rng = np.random.default_rng(int(scale * 7) + 3)
    n, nq = 3001, 257  # (odd: the ranges end on single points)
    sup = np.zeros((n, 4), np.float32)
    sup[:, :3] = rng.normal(0, 1.0, (n, 3)) * scale
    qry = np.zeros((nq, 4), np.float32)
    qry[:, :3] = rng.normal(0, 0.5, (nq, 3)) * scale
    # radii that ARE distances of the set (equality must count as outside), and their neighbours in fp32
    d = qry[0, None, :3] - sup[:, :3]
    r = d[:, 0] * d[:, 0]
    r = r + d[:, 1] * d[:, 1]
    r = r + d[:, 2] * d[:, 2]
    exact = np.sort(r)[[n // 7, n // 2, n - 3]]
    radii = [np.float32((0.4 * scale) ** 2)]
    for e in exact:
        radii += [e, np.nextafter(e, np.float32(0)), np.nextafter(e, np.float32(np.inf))]
    for r2 in radii:
        a, c = _device(sup, qry, r2)
        ref = _numpy(sup, qry, r2)
        assert np.array_equal(c, ref), f"plain compare vs numpy at r2 = {r2!r}"
        assert np.array_equal(a, ref), f"packed count vs numpy at r2 = {r2!r}: {int((a != ref).sum())} queries differ"


def test_far_points_and_no_points():
    sup = np.zeros((5, 4), np.float32)
    sup[:, 0] = [0.0, 1e19, -3e19, 3e38, 0.25]  # squares overflow: never inside
    qry = np.zeros((2, 4), np.float32)
    a, c = _device(sup, qry, 0.16)
    assert a.tolist() == c.tolist() == [2, 2]
    a, c = _device(sup[:0], qry, 0.16)
    assert a.tolist() == c.tolist() == [0, 0]
