"""Seeded differential fuzz on dense many-ring sensors (32 / 64 rings, 512-1024 azimuths): random scenes x random node
parameters, the C-ABI against the oracle.  These scans reach what the VLP-16 fuzz does not: the second run tier, the
workgroup ring tier, the large merge tiers, the long-list and whole-CU descriptor tiers (tools/fuzz_dense.py runs any seed
range and reports which tiers the cases used)."""
import ctypes as C

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


@pytest.mark.parametrize("block", range(2))
def test_dense_rows_through_the_one_small_launch(fx_hooks, oracle, block):
    """The dense tier's rows computed by dense_slow_loop, k_desc_mid's last workgroups (the list tier's body on scratch in HBM, what a batch gets whose
    predecessors had no dense row) instead of the tier's own kernels: forced by the test build's hook, the same scans, the
    same results — the choice between the two is the host's memory of earlier batches and may only ever decide speed."""
    fx_hooks(FX_DENSE_SLOW=1)
    checked, dense_rows = 0, 0
    for seed in range(7000 + 12 * block, 7000 + 12 * (block + 1)):
        s, p, roll, pitch, lim, what = dense_case(seed)
        ctx = capi.Context(p, lim)
        got = ctx.process_host([s], roll=roll, pitch=pitch)[0]
        h = (C.c_uint32 * 8)()
        ctx.lib.fx_debug_tier_hints.argtypes = [C.c_void_p, C.c_void_p]
        capi.check(ctx.lib.fx_debug_tier_hints(ctx.handle, h))
        dense_rows += int(h[4])
        ctx.close()
        if got["flags"]:
            continue
        util.compare_scan(got, oracle.run(p, s, roll=roll, pitch=pitch), tag=f"dense-slow seed {seed} {what}")
        checked += 1
    assert checked >= 10 and dense_rows > 0


def test_a_dense_row_after_sparse_batches_and_back(fxlib, oracle):
    """The product library's own choice: a context that has seen only VLP-16 scans (no dense row: the one small launch) gets
    a scan whose keypoints have thousands of support points — exact at once (dense_slow_loop computes it) —, then keeps the four
    kernels for the batches after it; every result equals the oracle's and a fresh context's."""
    s_d, p, roll, pitch, lim, what = None, None, 0.0, 0.0, None, None
    for seed in range(7000, 7100):  # the first case with dense rows and no capacity flag
        s, p, roll, pitch, lim, what = dense_case(seed)
        ctx = capi.Context(p, lim)
        got = ctx.process_host([s], roll=roll, pitch=pitch)[0]
        h = (C.c_uint32 * 8)()
        ctx.lib.fx_debug_tier_hints.argtypes = [C.c_void_p, C.c_void_p]
        capi.check(ctx.lib.fx_debug_tier_hints(ctx.handle, h))
        ctx.close()
        if h[4] > 0 and not got["flags"]:
            s_d, fresh = s, got
            break
    assert s_d is not None
    ora_d = oracle.run(p, s_d, roll=roll, pitch=pitch)
    sparse = s_d[::7].copy()  # a thinned scan: short support lists, no dense row
    ora_s = oracle.run(p, sparse, roll=roll, pitch=pitch)
    ctx = capi.Context(p, lim)
    seen = []
    for scan, ora in ((sparse, ora_s), (sparse, ora_s), (s_d, ora_d), (sparse, ora_s), (s_d, ora_d), (sparse, ora_s)):
        got = ctx.process_host([scan], roll=roll, pitch=pitch)[0]
        util.compare_scan(got, ora, tag=f"dense after sparse, batch {len(seen)}")
        h = (C.c_uint32 * 8)()
        capi.check(ctx.lib.fx_debug_tier_hints(ctx.handle, h))
        seen.append(int(h[4]))
        if scan is s_d:
            util.assert_bit_equal(got["descriptors"], fresh["descriptors"], "warm context == fresh context")
    ctx.close()
    assert seen[0] == seen[1] == seen[3] == seen[5] == 0 and seen[2] > 0 and seen[4] > 0, seen
