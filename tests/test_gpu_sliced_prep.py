"""Several workgroups a scan in the separate kernels' streaming pass and ring split (k_prep_count + k_prep_sliced + k_bucket_sliced:
batches that leave most of the chip idle with one workgroup a scan — 64 scans of 262 144 points).  Slice s streams the tiles
[s per, (s + 1) per) of the scan; its survivors land behind those of the slices before it (a counting pass first), its ring
counts start the next slice's places in every ring.  The results must be those of one workgroup a scan — and the oracle's —
for every slice count, ragged and empty scans, NaNs, window-boundary points and scans shorter than the slices."""
import numpy as np
import pytest

from feature_extraction_amd import capi
from tests import util

pytestmark = pytest.mark.gpu


def _scans():
    full = util.vlp16_scan(1000)
    nan = util.vlp16_scan(1001).copy()
    nan[::97, 0] = np.nan
    nan[5::131, 2] = np.inf
    rng = np.random.default_rng(9)
    pts = []
    for ring in range(16):  # points exactly on a ring window's edge: in two rings (ref: node.cpp:201)
        e = np.radians(-15.0 + 2.0 * ring + 1.0)
        az = np.radians(rng.uniform(-60, 60, 8))
        r = rng.uniform(8, 40, 8)
        pts.append(np.stack([r * np.cos(e) * np.cos(az), r * np.cos(e) * np.sin(az), r * np.sin(e), np.zeros(8)], 1))
    edges = np.concatenate(pts + [util.vlp16_scan(5)[:6000, :4].astype(np.float64)]).astype(np.float32)
    return [full, np.zeros((0, 4), np.float32), full[:1], full[:2049], nan, full[5000:9000], edges, full[::-1].copy(), util.vlp16_scan(1002)]


@pytest.mark.parametrize("slices", [2, 5, 16])
def test_vlp16_scans_through_the_sliced_kernels(fx_hooks, oracle, slices):
    fx_hooks(FX_FRONT=0, FX_PREP_SLICES=slices)
    scans = _scans()
    for preset in ("launch", "default"):
        p = capi.params(preset)
        ctx = capi.Context(p, capi.limits(len(scans), 28800, max_keypoints=512, max_total_keypoints=2048))
        for rep in range(2):  # (twice: the counts of the batch before must not leak into this one)
            got = ctx.process_host(scans if rep == 0 else scans[::-1], roll=0.02, pitch=-0.015)
            for b, s in enumerate(scans if rep == 0 else scans[::-1]):
                util.compare_scan(got[b], oracle.run(p, s, roll=0.02, pitch=-0.015), tag=f"{slices} slices {preset} rep {rep} scan {b}")
        ctx.close()


def test_the_product_library_slices_big_scans_in_small_batches(fxlib, oracle):
    """64 x 2048 and 128 x 2048 scans in batches of one to three (the product's own choice: sixteen workgroups a scan), bit-equal
    to the same scans through one workgroup a scan (the test build's hook) and to the oracle."""
    hd = [capi.synth_scan(capi.synth_cfg(10 + b, n_rings=64, n_az=2048, el0_deg=-24.8, el_step_deg=26.8 / 63, n_poles=256)) for b in range(3)]
    p = capi.params("launch", n_rings=64, el0_deg=-24.8, el_step_deg=26.8 / 63, secondary_max=64)
    lim = dict(max_candidates=3500, max_kpc_points=32768, max_total_keypoints=1024)
    ctx = capi.Context(p, capi.limits(3, 64 * 2048, **lim))
    got = ctx.process_host(hd, roll=0.02, pitch=-0.015)
    ctx.close()
    with capi.test_hooks():
        import os
        os.environ["FX_PREP_SLICES"] = "1"
        try:
            ctx = capi.Context(p, capi.limits(3, 64 * 2048, **lim))
            one = ctx.process_host(hd, roll=0.02, pitch=-0.015)
            ctx.close()
        finally:
            del os.environ["FX_PREP_SLICES"]
    for b in range(3):
        util.compare_scan(got[b], oracle.run(p, hd[b], roll=0.02, pitch=-0.015), tag=f"hdl64 sliced {b}")
        for key in ("filtered", "candidates", "kpc", "keypoints", "descriptors"):
            util.assert_bit_equal(got[b][key], one[b][key], f"sliced == one workgroup a scan: scan {b} {key}")


@pytest.mark.parametrize("batch", [1, 3, 8, 9])
def test_a_scan_per_call_takes_the_sliced_pass_with_the_clustering_kernel_behind_it(fxlib, fx_hooks, oracle, batch):
    """VERDICT r5 #5 (ref: node.cpp:72, :386 — the reference handles one scan per callback): batches of up to eight VLP-16 scans
    take k_prep_count + k_prep_sliced + k_bucket_sliced (sixteen workgroups a scan) with k_front_cd behind them instead of
    k_front; from nine on the fused kernel.  The product's own choice against the oracle, and bit-identical to the other path
    forced by the test build's hook — ragged, empty, NaN, window-boundary and reversed scans included, twice per context."""
    scans = (_scans() * 2)[:batch]
    p = capi.params("launch")
    lim = capi.limits(batch, 28800, max_keypoints=512, max_total_keypoints=4096)
    ctx = capi.Context(p, lim)
    got = [ctx.process_host(scans if rep == 0 else scans[::-1], roll=0.02, pitch=-0.015) for rep in range(2)]
    ctx.close()
    for b, s in enumerate(scans):
        util.compare_scan(got[0][b], oracle.run(p, s, roll=0.02, pitch=-0.015), tag=f"batch of {batch}, scan {b}")
    fx_hooks(FX_FRONT_STREAM=0 if batch <= 8 else 1)  # the other path
    ctx = capi.Context(p, lim)
    other = ctx.process_host(scans, roll=0.02, pitch=-0.015)
    ctx.close()
    for b in range(batch):
        for key in ("filtered", "candidates", "cand_size", "kpc", "kpc_cand", "cand_keypoint", "keypoints", "kp_size", "kp_neighbors", "descriptors"):
            util.assert_bit_equal(got[0][b][key], other[b][key], f"batch of {batch}: sliced + k_front_cd against k_front, scan {b} {key}")
            util.assert_bit_equal(got[1][batch - 1 - b][key], other[b][key], f"batch of {batch}: second batch of the context, scan {b} {key}")
