"""GPU parity: the HIP path through the C-ABI vs the CPU oracle on identical inputs."""
import numpy as np
import pytest

from feature_extraction_amd import capi
from tests import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("preset", ["default", "launch"])
@pytest.mark.parametrize("leveled", [False, True])
def test_vlp16_batch_matches_oracle(fxlib, oracle, preset, leveled):
    B = 6
    scans = [util.vlp16_scan(1000 + b) for b in range(B)]
    roll, pitch = (0.02, -0.015) if leveled else (0.0, 0.0)
    p = capi.params(preset)
    ctx = capi.Context(p, capi.limits(B, 28800))
    got = ctx.process_host(scans, roll=roll, pitch=pitch)
    total_k = 0
    worst = 0.0
    inexact = 0
    for b in range(B):
        ora = oracle.run(p, scans[b], roll=roll, pitch=pitch)
        st = util.compare_scan(got[b], ora, tag=f"{preset} scan {b}")
        total_k += st["K"]
        worst = max(worst, st["max_abs"])
        inexact += st["n_inexact"]
    assert total_k > 0
    print(f"\n[{preset} leveled={leveled}] K total {total_k}, descriptor max|diff| {worst:.3g}, inexact values {inexact}")
    ctx.close()
