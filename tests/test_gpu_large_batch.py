"""BASELINE config 2 at full size (1024 x 16 x 1800) through size-independent properties:
determinism, sampled parity, batch composition independence, offsets consistency."""
import numpy as np
import pytest

from feature_extraction_amd import capi
from tests import util

pytestmark = pytest.mark.gpu


def test_batch_1024_properties(fxlib, oracle):
    import torch
    B, N = 1024, 28800
    uniq = [util.vlp16_scan(1000 + b) for b in range(64)]
    order = np.arange(B) % 64
    host = np.stack([uniq[i] for i in order])
    d = torch.from_numpy(host).cuda()
    p = capi.params("launch")
    ctx = capi.Context(p, capi.limits(B, N))
    descs = ctx.make_descs([d.data_ptr() + b * N * 16 for b in range(B)], [N] * B, 16, 0.02, -0.015)
    flags = capi.FX_IN_DEVICE | capi.FX_OUT_HOST
    v = ctx.process_raw(descs, B, flags)
    res = ctx.unpack(v, debug=False)
    assert all(r["flags"] == 0 for r in res)
    # a scan's result does not depend on its position in the batch or on its neighbours
    for b in range(64, B):
        a, c = res[b], res[b - 64]
        assert a["n_keypoints"] == c["n_keypoints"]
        util.assert_bit_equal(a["keypoints"], c["keypoints"], f"keypoints {b}")
        util.assert_bit_equal(a["descriptors"], c["descriptors"], f"descriptors {b}")
    # offsets are the exclusive prefix of the counts
    n_kp = np.array([r["n_keypoints"] for r in res])
    assert v.total_keypoints == n_kp.sum() > 40 * B
    # sampled parity against the oracle
    small = capi.Context(p, capi.limits(8, N))
    for b in (0, 17, 63):
        got = small.process_host([uniq[b]], roll=0.02, pitch=-0.015)[0]
        util.compare_scan(got, oracle.run(p, uniq[b], roll=0.02, pitch=-0.015), tag=f"sample {b}")
        util.assert_bit_equal(got["keypoints"], res[b]["keypoints"], "batch of 1 == batch of 1024")
        util.assert_bit_equal(got["descriptors"], res[b]["descriptors"], "batch of 1 == batch of 1024")
    # and running it again gives the same bits
    res2 = ctx.unpack(ctx.process_raw(descs, B, flags), debug=False)
    for a, c in zip(res, res2):
        util.assert_bit_equal(a["descriptors"], c["descriptors"], "repeatability")
    small.close()
    ctx.close()
