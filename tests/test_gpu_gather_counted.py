"""The support gather by several workgroups a scan in two launches (k_gather_count: every slice counts its hits per keypoint;
k_gather_scatter: every slice writes its hits straight to their list slots, its first position per keypoint being the sum of
the earlier slices' counts; ref: node.cpp:343-353 — the radius searches of pcl::ShapeContext3DEstimation): what batches of few
big scans take (64 scans of 262 144 points are 64 workgroups on 256 CUs).  Against one workgroup a scan: the same support sets
(list order is free: the descriptor kernels sort by unique keys), so the same neighbour counts and bit-identical descriptors —
with lists that overflow into the scan's region, with empty and ragged scans, with more slices than tiles."""
import numpy as np
import pytest

from feature_extraction_amd import capi
from tests import util

pytestmark = pytest.mark.gpu
KEYS = ("filtered", "candidates", "keypoints", "kp_size", "kp_neighbors", "descriptors")


def _run(fx_hooks, counted, p, lim, scans, **kw):
    fx_hooks(FX_GATHER_COUNTED=counted)
    ctx = capi.Context(p, lim)
    out = [ctx.process_host(scans, **kw) for _ in range(2)]  # (twice: the rows keep their content between batches)
    ctx.close()
    return out


def test_dense_scan_with_overflowing_lists(fx_hooks, oracle):
    """128 rings x 2048, R = 2 m: rows of thousands of support points — beyond the 1024 list slots given here, so most of a
    row lives in the scan's overflow region, whose positions the counted slices derive from the same counts."""
    cfg = dict(n_rings=128, n_az=2048, el0_deg=-25.0, el_step_deg=40.0 / 127, n_poles=256)
    p = capi.params("launch", n_rings=128, el0_deg=-25.0, el_step_deg=40.0 / 127, secondary_max=128, descriptor_radius=2.0)
    scans = [capi.synth_scan(capi.synth_cfg(10 + b, **cfg)) for b in range(2)] + [np.zeros((0, 4), np.float32)]
    scans.append(scans[0][:150000].copy())
    lim = capi.limits(len(scans), 128 * 2048, max_candidates=8192, max_kpc_points=65536, max_keypoints=512, max_total_keypoints=2048, max_neighbors=1024)
    ref = _run(fx_hooks, 1, p, lim, scans, roll=0.02, pitch=-0.015)
    assert all(g["flags"] == 0 for g in ref[0]) and int(max(g["kp_neighbors"].max() for g in ref[0] if g["n_keypoints"])) > 4000
    util.compare_scan(ref[0][0], oracle.run(p, scans[0], roll=0.02, pitch=-0.015), tag="one workgroup a scan")
    for counted in (5, 12, 16):
        got = _run(fx_hooks, counted, p, lim, scans, roll=0.02, pitch=-0.015)
        for rep in range(2):
            for b in range(len(scans)):
                assert got[rep][b]["flags"] == 0
                for key in KEYS:
                    util.assert_bit_equal(got[rep][b][key], ref[0][b][key], f"{counted} slices, run {rep}, scan {b}: {key}")


def test_vlp16_scans_and_more_slices_than_tiles(fx_hooks, oracle):
    """Small scans (a slice of the sixteen may get no tile at all), window-boundary and NaN points, a keypoint without neighbours
    (the RNG ordinals are k_rng_ord's when several workgroups share a scan)."""
    scans = [util.vlp16_scan(1000 + b) for b in range(5)]
    scans[1][::7, :3] = np.nan
    scans[2] = scans[2][:3000].copy()
    scans.append(np.zeros((0, 4), np.float32))
    p = capi.params("launch", descriptor_radius=0.3)  # (a small radius: several keypoints have no neighbour)
    lim = capi.limits(len(scans), 28800)
    ref = _run(fx_hooks, 1, p, lim, scans, roll=0.01, pitch=0.02)
    assert any((g["kp_neighbors"] == 0).any() for g in ref[0] if g["n_keypoints"])
    for b in (0, 1, 2):
        util.compare_scan(ref[0][b], oracle.run(p, scans[b], roll=0.01, pitch=0.02), tag=f"one workgroup a scan, scan {b}")
    for counted in (3, 16):
        got = _run(fx_hooks, counted, p, lim, scans, roll=0.01, pitch=0.02)
        for b in range(len(scans)):
            for key in KEYS:
                util.assert_bit_equal(got[1][b][key], ref[0][b][key], f"{counted} slices, scan {b}: {key}")
