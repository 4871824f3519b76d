"""BASELINE config 2 at full size (1024 x 16 x 1800), both presets: every distinct scan of the batch against the
oracle, plus size-independent properties — determinism, batch composition independence, offsets consistency."""
import concurrent.futures as cf
import os

import numpy as np
import pytest

from feature_extraction_amd import capi
from tests import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("preset", ["launch", "default"])
def test_batch_1024_properties(fxlib, oracle, preset):
    import torch
    B, N, U = 1024, 28800, 64
    uniq = [util.vlp16_scan(1000 + b) for b in range(U)]
    order = np.arange(B) % U
    host = np.stack([uniq[i] for i in order])
    d = torch.from_numpy(host).cuda()
    p = capi.params(preset)
    ctx = capi.Context(p, capi.limits(B, N))
    descs = ctx.make_descs([d.data_ptr() + b * N * 16 for b in range(B)], [N] * B, 16, 0.02, -0.015)
    flags = capi.FX_IN_DEVICE | capi.FX_OUT_HOST
    v = ctx.process_raw(descs, B, flags)
    res = ctx.unpack(v, debug=False)
    assert all(r["flags"] == 0 for r in res)
    # a scan's result does not depend on its position in the batch or on its neighbours
    for b in range(U, B):
        a, c = res[b], res[b - U]
        assert a["n_keypoints"] == c["n_keypoints"]
        util.assert_bit_equal(a["keypoints"], c["keypoints"], f"keypoints {b}")
        util.assert_bit_equal(a["descriptors"], c["descriptors"], f"descriptors {b}")
    # offsets are the exclusive prefix of the counts
    n_kp = np.array([r["n_keypoints"] for r in res])
    assert v.total_keypoints == n_kp.sum() > (40 if preset == "launch" else 1) * B
    # every distinct scan of the batch against the oracle (kd-tree search, one scan per host thread), with the
    # membership arrays of a second, debug pass over the first U scans
    full = ctx.unpack(ctx.process_raw(descs, U, flags | capi.FX_OUT_CLOUDS | capi.FX_OUT_DEBUG))
    with cf.ThreadPoolExecutor(max_workers=min(U, os.cpu_count() or 1)) as ex:
        oras = list(ex.map(lambda s: oracle.run(p, s, roll=0.02, pitch=-0.015), uniq))
    for b in range(U):
        util.compare_scan(full[b], oras[b], tag=f"{preset} scan {b}")
        util.assert_bit_equal(full[b]["keypoints"], res[b]["keypoints"], "batch of 64 == batch of 1024")
        util.assert_bit_equal(full[b]["descriptors"], res[b]["descriptors"], "batch of 64 == batch of 1024")
    # and running it again gives the same bits
    res2 = ctx.unpack(ctx.process_raw(descs, B, flags), debug=False)
    for a, c in zip(res, res2):
        util.assert_bit_equal(a["descriptors"], c["descriptors"], "repeatability")
    ctx.close()
