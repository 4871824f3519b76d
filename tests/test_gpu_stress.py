"""Race hunt: the clustering kernels use lock-free union-find in LDS; a lost update shows up as a
rare run-to-run difference.  The same batch goes through one context many times and every output
must stay bit-identical (and match the oracle)."""
import numpy as np
import pytest

from feature_extraction_amd import capi
from tests import util

pytestmark = pytest.mark.gpu
KEYS = ("filtered", "candidates", "cand_size", "cand_keypoint", "kpc", "kpc_cand", "keypoints", "kp_size",
        "kp_neighbors", "descriptors")


@pytest.mark.parametrize("preset", ["launch", "default"])
def test_repeated_batches_are_bit_identical(fxlib, oracle, preset):
    B, reps = 48, 25
    rng = np.random.default_rng(0)
    scans = [util.vlp16_scan(1000 + b) for b in range(B)]
    scans[3][rng.choice(28800, 300, replace=False), 0] = np.nan
    scans[5] = scans[5][rng.permutation(28800)]           # unordered input: all-pairs clustering path
    scans[7] = util.vlp16_scan(7, n_poles=256)            # many clusters: order replay beyond 16
    p = capi.params(preset)
    ctx = capi.Context(p, capi.limits(B, 28800, max_total_keypoints=B * 200))
    ref = ctx.process_host(scans, roll=0.02, pitch=-0.015)
    for b in (0, 3, 5, 7):
        util.compare_scan(ref[b], oracle.run(p, scans[b], roll=0.02, pitch=-0.015), tag=f"{preset} scan {b}")
    for rep in range(reps):
        got = ctx.process_host(scans, roll=0.02, pitch=-0.015)
        for b in range(B):
            for k in KEYS:
                util.assert_bit_equal(got[b][k], ref[b][k], f"{preset} rep {rep} scan {b} {k}")
    ctx.close()
