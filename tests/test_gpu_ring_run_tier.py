"""The run tier of the ring stage (k_rings_runs: one wavefront per ring, run / segment tables in LDS, the ring's
first 170 points cached, the first 256 kept in registers for the member copy) at its edges, against the oracle:
long rings, rings at and past the 128-run table, many near run pairs (edges between non-consecutive runs), members
beyond the cached and the register-held points, and the hand-over to the workgroup tiers."""
import ctypes as C

import numpy as np
import pytest

from feature_extraction_amd import capi
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def separate_kernels(fx_hooks):
    """These cases are about k_rings_runs / k_rings_runs2 / k_rings_large: the test build's hook takes the separate
    kernels instead of the fused front kernel (tests/test_gpu_front.py runs the same shapes through that one)."""
    fx_hooks(FX_FRONT=0)


def ring_points(ring, az_deg, rng_m):
    """Returns of VLP-16 ring `ring` (elevation -15 + 2 ring degrees) at the given azimuths / ranges, in that order."""
    el = np.radians(-15.0 + 2.0 * ring)
    az = np.radians(np.asarray(az_deg, np.float64))
    r = np.asarray(rng_m, np.float64)
    return np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el), np.zeros_like(r)], 1).astype(np.float32)


def pole(ring, az0, rng_m, k=5):
    return ring_points(ring, az0 + 0.13 * np.arange(k), np.full(k, rng_m))


def long_ring_with_late_poles():
    """Ring 8: a 600-point arc (one run, fails the gate), then poles whose members sit beyond point 256."""
    arc = ring_points(8, np.linspace(-40, 40, 600), np.full(600, 10.0))
    poles = [pole(8, az, 25.0) for az in (-60, -50, 50, 60, 70)]
    return np.concatenate([arc] + poles)


def isolated_points(ring, count):
    """`count` single-point runs on one ring: consecutive returns alternate between two ranges 15 m apart."""
    az = np.linspace(-88, 88, count)
    return ring_points(ring, az, np.where(np.arange(count) % 2 == 0, 30.0, 45.0))


def interleaved_arc(ring, count, step_m=0.3, rng_m=20.0):
    """Points 0.3 m apart along an arc, visited 0, h, 1, h + 1, ...: every point is its own run and near three runs
    on either side, none of them its neighbour in input order."""
    h = count // 2
    order = np.stack([np.arange(h), h + np.arange(h)], 1).reshape(-1)
    az = np.degrees(step_m / rng_m) * order - 20.0
    return ring_points(ring, az, np.full(len(order), rng_m))


def two_arcs_in_blocks(ring):
    """200 points: arcs A (20 m) and B (20.9 m) visited in blocks of ten — runs of twenty points joined across the
    blocks into one cluster — then two poles whose points lie beyond the 170 cached ones."""
    blocks = []
    for k in range(9):
        az = -30.0 + 3.0 * k + 0.3 * np.arange(10)
        blocks += [ring_points(ring, az, np.full(10, 20.0)), ring_points(ring, az, np.full(10, 20.9))]
    return np.concatenate(blocks + [pole(ring, 40.0, 30.0), pole(ring, 50.0, 24.0, k=7)])


def _run(oracle, scans, tag, p=None, **lim_over):
    p = p or capi.params("launch")
    ctx = capi.Context(p, capi.limits(len(scans), 28800, **lim_over))
    got = ctx.process_host(scans)
    cnt = (C.c_uint32 * 16)()
    ctx.lib.fx_debug_counters.argtypes = [C.c_void_p, C.c_void_p]
    capi.check(ctx.lib.fx_debug_counters(ctx.handle, cnt))
    for b, s in enumerate(scans):
        util.compare_scan(got[b], oracle.run(p, s), tag=f"{tag} scan {b}")
    ctx.close()
    return got, {"second": cnt[0], "workgroup": cnt[5]}


def test_long_ring_and_members_beyond_the_register_held_points(fxlib, oracle):
    got, tiers = _run(oracle, [long_ring_with_late_poles()], "long ring")
    assert got[0]["flags"] == 0 and tiers == {"second": 0, "workgroup": 0}
    assert len(got[0]["candidates"]) == 5  # the poles; the arc fails the diameter gate


def test_run_table_boundary(fxlib, oracle):
    """128 single-point runs (= 128 clusters) stay in the run tier; 129 go to the workgroup tier."""
    got, tiers = _run(oracle, [isolated_points(9, 128), isolated_points(9, 129),
                               np.concatenate([isolated_points(9, 128), isolated_points(10, 129), isolated_points(11, 127)])],
                      "run table", max_ring_candidates=512, max_keypoints=512, max_total_keypoints=4096)
    assert [g["flags"] for g in got] == [0, 0, 0]
    assert tiers["workgroup"] == 2
    assert [len(g["candidates"]) for g in got][:2] == [128, 129]


def test_edges_between_non_consecutive_runs(fxlib, oracle):
    """30 interleaved runs (about 90 near pairs) merge into one cluster in the run tier; 100 of them (about 300
    near pairs, past the pair list) are handed over."""
    got, tiers = _run(oracle, [interleaved_arc(6, 30), interleaved_arc(6, 100), np.concatenate([interleaved_arc(5, 40), interleaved_arc(6, 30)])],
                      "interleaved")
    assert [g["flags"] for g in got] == [0, 0, 0]
    assert tiers["workgroup"] >= 1


def test_members_beyond_the_cached_points(fxlib, oracle):
    got, tiers = _run(oracle, [two_arcs_in_blocks(7)], "two arcs")
    assert got[0]["flags"] == 0 and tiers == {"second": 0, "workgroup": 0}
    assert len(got[0]["candidates"]) == 2


def test_all_shapes_in_one_batch(fxlib, oracle):
    scans = [long_ring_with_late_poles(), two_arcs_in_blocks(3), interleaved_arc(12, 60), isolated_points(14, 100),
             np.concatenate([long_ring_with_late_poles(), two_arcs_in_blocks(9), interleaved_arc(6, 36), isolated_points(10, 128)])]
    got, _ = _run(oracle, scans, "mixed", max_ring_candidates=512, max_keypoints=512, max_total_keypoints=4096)
    assert [g["flags"] for g in got] == [0] * len(scans)


def test_second_run_tier_of_many_ring_sensors(fxlib, oracle):
    """Sensors of more than 16 rings get a second run tier (384 segments, 256 runs and clusters) behind the first: 200
    single-point runs stay there, 257 need the workgroup tier; an interleaved arc beyond the near-pair list too."""
    p = capi.params("launch", n_rings=32, el0_deg=-15.0, el_step_deg=1.0, secondary_max=32, cluster_tolerance=0.5)
    scans = [isolated_points(9, 200), isolated_points(9, 256), isolated_points(9, 257), interleaved_arc(6, 100, step_m=0.15),
             np.concatenate([isolated_points(7, 180), long_ring_with_late_poles(), two_arcs_in_blocks(9)])]
    got, tiers = _run(oracle, scans, "second run tier", p=p, max_ring_candidates=512, max_keypoints=512, max_total_keypoints=4096)
    assert [g["flags"] for g in got] == [0] * len(scans)
    assert [len(g["candidates"]) for g in got][:3] == [200, 256, 257]
    assert tiers["second"] >= 5 and 1 <= tiers["workgroup"] <= 3
