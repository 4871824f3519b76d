"""Seeded differential fuzz on dense many-ring sensors (32 / 64 rings, 512-1024 azimuths): random scenes x random node
parameters, the C-ABI against the oracle.  These scans reach what the VLP-16 fuzz does not: the second run tier, the
workgroup ring tier, the large merge tiers, the long-list and whole-CU descriptor tiers (tools/fuzz_dense.py runs any seed
range and reports which tiers the cases used)."""
import numpy as np
import pytest

from feature_extraction_amd import capi
from tests import util

pytestmark = pytest.mark.gpu


def dense_case(seed, n_az_override=None, rings_override=None):
    rng = np.random.default_rng(seed)
    R = int(rng.choice([32, 64]))
    n_az = int(rng.choice([512, 768, 1024]))
    if rings_override:
        R = int(rings_override)
    if n_az_override:
        n_az = int(n_az_override)
    el_span = float(rng.uniform(20.0, 40.0))
    el0 = -float(rng.uniform(12.0, 25.0))
    cfg = capi.synth_cfg(int(rng.integers(1, 1 << 30)), n_rings=R, n_az=n_az, el0_deg=el0, el_step_deg=el_span / (R - 1),
                         n_poles=int(rng.integers(0, 200)), pole_radius=float(rng.uniform(0.03, 0.4)),
                         sensor_height=float(rng.uniform(1.0, 2.5)), wall_radius=float(rng.uniform(30, 100)))
    s = capi.synth_scan(cfg)
    over = dict(n_rings=R, el0_deg=el0, el_step_deg=el_span / (R - 1), secondary_max=R,
                cluster_tolerance=float(rng.uniform(0.15, 1.2)), cluster_min_count=int(rng.integers(1, 6)),
                cluster_max_count=int(rng.integers(20, 1500)), cluster_radius_threshold=float(rng.uniform(0.08, 0.5)),
                number_detection_channels=int(rng.integers(1, 6)), descriptor_radius=float(rng.uniform(0.5, 3.0)),
                cloud_leveling=int(rng.integers(0, 2)))
    p = capi.params(str(rng.choice(["default", "launch"])), **over)
    if rng.integers(0, 4) == 0:  # azimuth blocks out of order
        blocks = np.array_split(np.arange(len(s)), 24)
        s = np.ascontiguousarray(s[np.concatenate([blocks[i] for i in rng.permutation(len(blocks))])])
    roll, pitch = float(rng.uniform(-0.05, 0.05)), float(rng.uniform(-0.05, 0.05))
    lim = capi.limits(1, len(s), max_candidates=8192, max_kpc_points=2 * len(s), max_keypoints=1024, max_total_keypoints=1024,
                      max_ring_candidates=1024)
    return s, p, roll, pitch, lim, dict(R=R, n_az=n_az, over=over)


@pytest.mark.parametrize("block", range(4))
def test_dense_many_ring_scans(fxlib, oracle, block):
    checked = 0
    for seed in range(7000 + 12 * block, 7000 + 12 * (block + 1)):
        s, p, roll, pitch, lim, what = dense_case(seed)
        ctx = capi.Context(p, lim)
        got = ctx.process_host([s], roll=roll, pitch=pitch)[0]
        ctx.close()
        if got["flags"]:  # a capacity flag (never silent) is a legitimate outcome of random parameters
            continue
        util.compare_scan(got, oracle.run(p, s, roll=roll, pitch=pitch), tag=f"dense seed {seed} {what}")
        checked += 1
    assert checked >= 10
