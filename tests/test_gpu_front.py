"""The fused front kernel (k_front: filter, ring split, per-ring clustering and secondary merge of a scan in ONE launch, the
intermediates in LDS; ref: node.cpp:147-259) and the two kernels behind it: k_front_redo (the general kernels' bodies in
k_front's shape, for scans that do not fit its tables) and k_slow (the same bodies on scratch in HBM, for rings and merges
that do not fit that shape's LDS either).  All three are launched with every batch: what a scan gets never depends on what
earlier batches needed.  Which of them took a scan is read from the context's tier hints (6: scans handed to k_front_redo,
7: to k_slow); every result is compared with the oracle."""
import ctypes as C

import numpy as np
import pytest

from feature_extraction_amd import capi
from tests import util
from tests.test_gpu_ring_run_tier import (interleaved_arc, isolated_points, long_ring_with_late_poles, ring_points,
                                          two_arcs_in_blocks)

pytestmark = pytest.mark.gpu

FRONT_CAP, FRONT_RUNS, FRONT_PAIRS = 3328, 512, 512  # FX_FRONT_CAP / _RUNS / _PAIRS of csrc/fx_kernels.hip


def _hints(ctx):
    h = (C.c_uint32 * 8)()
    ctx.lib.fx_debug_tier_hints.argtypes = [C.c_void_p, C.c_void_p]
    capi.check(ctx.lib.fx_debug_tier_hints(ctx.handle, h))
    return int(h[6]), int(h[7])


def _run(oracle, scans, tag, p=None, roll=0.0, pitch=0.0, ctx=None, **lim_over):
    p = p or capi.params("launch")
    own = ctx is None
    if own:
        ctx = capi.Context(p, capi.limits(len(scans), 28800, **lim_over))
    got = ctx.process_host(scans, roll=roll, pitch=pitch)
    hints = _hints(ctx)
    for b, s in enumerate(scans):
        util.compare_scan(got[b], oracle.run(p, s, roll=roll, pitch=pitch), tag=f"{tag} scan {b}")
    if own:
        ctx.close()
    return got, hints


def test_ring_shapes_stay_in_the_front_kernel(fxlib, oracle):
    """The run tier's edge cases (tests/test_gpu_ring_run_tier.py) through k_front: long rings, many single-point runs, edges
    between non-consecutive runs, members far into a ring — none of them needs another kernel."""
    scans = [long_ring_with_late_poles(), two_arcs_in_blocks(3), interleaved_arc(12, 60), isolated_points(14, 100), isolated_points(9, 129),
             interleaved_arc(6, 100),
             np.concatenate([long_ring_with_late_poles(), two_arcs_in_blocks(9), interleaved_arc(6, 36), isolated_points(10, 128)]),
             np.zeros((0, 4), np.float32)]
    got, hints = _run(oracle, scans, "front shapes", max_ring_candidates=512, max_keypoints=512, max_total_keypoints=4096)
    assert [g["flags"] for g in got] == [0] * len(scans) and hints == (0, 0)
    assert len(got[4]["candidates"]) == 129 and len(got[0]["candidates"]) == 5


def test_bench_scans_stay_in_the_front_kernel(fxlib, oracle):
    scans = [util.vlp16_scan(1000 + b) for b in range(12)]
    for preset in ("launch", "default"):
        got, hints = _run(oracle, scans, f"front {preset}", p=capi.params(preset), roll=0.02, pitch=-0.015)
        assert hints == (0, 0) and sum(g["n_keypoints"] for g in got) > 0


def test_run_table_at_its_limit(fxlib, oracle):
    """FX_FRONT_RUNS single-point runs over four rings fit k_front's table; one more hands the scan to k_front_redo."""
    def runs(total):  # (rings 7 .. 10: the ones whose returns at 30 / 45 m pass the z limits of the filter)
        per, rest = divmod(total, 4)
        return np.concatenate([isolated_points(7 + r, per + (1 if r < rest else 0)) for r in range(4)])
    lim = dict(max_ring_candidates=512, max_keypoints=512, max_total_keypoints=4096)
    got, hints = _run(oracle, [runs(FRONT_RUNS)], "runs at the limit", **lim)
    assert hints == (0, 0) and got[0]["flags"] == 0 and len(got[0]["candidates"]) == FRONT_RUNS
    got, hints = _run(oracle, [runs(FRONT_RUNS + 1), runs(FRONT_RUNS)], "runs past the limit", **lim)
    assert hints == (1, 0) and [g["flags"] for g in got] == [0, 0] and len(got[0]["candidates"]) == FRONT_RUNS + 1


def test_near_pair_list_past_its_limit(fxlib, oracle):
    """Interleaved arcs: every run is near three runs on either side.  Three rings of 100 such runs have about 900 near run
    pairs — more than k_front lists —, one has about 300."""
    one = interleaved_arc(6, 100)
    three = np.concatenate([interleaved_arc(6, 100), interleaved_arc(7, 100), interleaved_arc(8, 100)])
    got, hints = _run(oracle, [one, three], "near pairs")
    assert hints == (1, 0) and [g["flags"] for g in got] == [0, 0]


def test_more_ring_entries_than_the_front_kernel_holds(fxlib, oracle):
    """A filter box all around the sensor: ~5000 survivors a scan (k_front holds FX_FRONT_CAP).  The first batch goes through
    k_front_redo scan by scan; a context that keeps getting such scans goes back to the separate kernels after it."""
    p = capi.params("launch", x_min=-100.0)
    scans = [util.vlp16_scan(1000 + b) for b in range(4)]
    ctx = capi.Context(p, capi.limits(4, 28800))
    seen = []
    for rep in range(3):
        got, hints = _run(oracle, scans, f"full circle rep {rep}", p=p, roll=0.02, pitch=-0.015, ctx=ctx)
        assert all(len(g["filtered"]) > FRONT_CAP and g["flags"] == 0 for g in got)
        seen.append(hints)
    ctx.close()
    assert seen[0] == (4, 0) and seen[1] == (0, 0) and seen[2] == (0, 0), seen  # (nothing handed on: k_front did not run)


@pytest.mark.parametrize("force", [1, 2])
def test_every_scan_through_the_kernels_behind_the_front_kernel(fx_hooks, oracle, force):
    """The test build's hook hands every scan to k_front_redo (1), and from there every ring and merge to k_slow (2)."""
    fx_hooks(FX_FRONT_FORCE=force)
    scans = [util.vlp16_scan(1000 + b) for b in range(6)] + [np.zeros((0, 4), np.float32), long_ring_with_late_poles(), interleaved_arc(6, 100)]
    for preset in ("launch", "default"):
        got, hints = _run(oracle, scans, f"forced {force} {preset}", p=capi.params(preset), roll=0.02, pitch=-0.015)
        assert [g["flags"] for g in got] == [0] * len(scans)
        assert hints == (len(scans) - 1, len(scans) - 1 if force == 2 else 0)  # (the empty scan never gets that far)


def shells_ring(ring=9, cluster_pts=4):
    """A ring of ~1700 points in ~420 clusters (six range shells, visited in azimuth order): more points than the small ring
    tier of k_front_redo's LDS image holds, more clusters than its larger one orders."""
    az, rg = [], []
    for shell in (20.0, 25.0, 30.0, 35.0, 40.0, 45.0):
        step = np.degrees(1.45 / shell)  # cluster centres 1.45 m apart along the shell (the tolerance is 1 m)
        for c in np.arange(-86.0, 86.0, step):
            az += list(c + 0.1 * np.arange(cluster_pts))
            rg += [shell] * cluster_pts
    order = np.argsort(np.array(az), kind="stable")
    return ring_points(ring, np.array(az)[order], np.array(rg)[order])


BIG_LIM = dict(max_ring_candidates=2048, max_candidates=4096, max_keypoints=512, max_total_keypoints=1024, max_kpc_points=8192)


def test_a_scan_beyond_the_lds_tiers_is_exact_on_first_presentation(fxlib, oracle):
    """VERDICT r4 #1 (ref: node.cpp:72-145 never drops a scan): a ring that fits neither k_front's tables nor k_front_redo's
    LDS image is k_slow's work — and k_slow goes with every batch, so the scan is exact (flags 0, oracle-equal) the FIRST time
    a warm context sees it, whatever came before; the batches around it are untouched."""
    shells = shells_ring()
    assert len(shells) > 1600
    big = np.concatenate([shells, long_ring_with_late_poles()])
    small = util.vlp16_scan(1000)
    p = capi.params("launch")
    got, hints = _run(oracle, [big], "slow tier, fresh context", p=p, **BIG_LIM)
    assert hints == (1, 1) and got[0]["flags"] == 0 and len(got[0]["candidates"]) > 400
    ctx = capi.Context(p, capi.limits(1, 28800, **BIG_LIM))
    for rep in range(2):
        _, hints = _run(oracle, [small], f"slow tier, small before {rep}", p=p, ctx=ctx)
        assert hints == (0, 0)
    got, hints = _run(oracle, [big], "slow tier, first presentation in a warm context", p=p, ctx=ctx)
    assert hints == (1, 1) and got[0]["flags"] == 0 and len(got[0]["candidates"]) > 400
    _, hints = _run(oracle, [small], "slow tier, small again", p=p, ctx=ctx)
    assert hints == (0, 0)
    got, hints = _run(oracle, [big], "slow tier, again", p=p, ctx=ctx)
    assert hints == (1, 1) and got[0]["flags"] == 0
    ctx.close()


def test_any_interleaving_of_small_and_big_scans_gives_the_same_results(fxlib, oracle):
    """Per-scan results are a function of the scan alone: the same scans in different orders, batch compositions and contexts
    (k_front only / k_front_redo / k_slow needed or not by the batch before) come back bit-identical."""
    p = capi.params("launch")
    big = np.concatenate([shells_ring(), long_ring_with_late_poles()])
    pool = {"small0": util.vlp16_scan(1000), "small1": util.vlp16_scan(1001), "big": big, "runs": np.concatenate([isolated_points(7 + r, 129) for r in range(4)]),
            "pairs": np.concatenate([interleaved_arc(6, 100), interleaved_arc(7, 100), interleaved_arc(8, 100)]), "empty": np.zeros((0, 4), np.float32)}
    lim = dict(BIG_LIM, max_total_keypoints=4096)
    ref = {}
    ctx = capi.Context(p, capi.limits(4, 28800, **lim))
    for name, s in pool.items():  # every scan alone, in a context that has seen nothing but small scans
        ctx.process_host([pool["small0"]])
        ref[name] = ctx.process_host([s])[0]
        assert ref[name]["flags"] == 0, name
    util.compare_scan(ref["big"], oracle.run(p, big), tag="big alone")
    util.compare_scan(ref["runs"], oracle.run(p, pool["runs"]), tag="runs alone")
    rng = np.random.default_rng(5)
    names = list(pool)
    for trial in range(12):
        batch = [names[i] for i in rng.integers(0, len(names), int(rng.integers(1, 5)))]
        got = ctx.process_host([pool[n] for n in batch])
        for n, g in zip(batch, got):
            for key in ("flags", "n_keypoints"):
                assert g[key] == ref[n][key], (trial, batch, n, key)
            for key in ("filtered", "candidates", "cand_size", "kpc", "kpc_cand", "cand_keypoint", "keypoints", "kp_size", "kp_neighbors", "descriptors"):
                util.assert_bit_equal(g[key], ref[n][key], f"trial {trial} {batch} {n} {key}")
    ctx.close()


def test_rings_beyond_the_lds_ceiling(fxlib, fx_hooks, oracle):
    """VERDICT r4 #7 (ref: node.cpp:273-274: the reference's only limit is cluster_max_count): a ring of more points than any
    workgroup's LDS holds (2400 until round 4) with too many runs for the run tiers — a 0.1-degree sensor's 3600 returns in
    many small clusters — is clustered on scratch in HBM (k_slow), unflagged and oracle-equal: through the fused front path
    and through the separate kernels."""
    az, rg = [], []
    for shell in (12.0, 15.0, 18.0, 21.0, 24.0, 27.0, 30.0, 33.0, 36.0, 39.0, 42.0, 45.0):
        step = np.degrees(1.45 / shell)
        for c in np.arange(-88.0, 88.0, step):
            az += list(c + 0.05 * np.arange(5))
            rg += [shell] * 5
    order = np.argsort(np.array(az), kind="stable")
    ring = ring_points(9, np.array(az)[order], np.array(rg)[order])
    assert len(ring) > 3000
    scan = np.concatenate([ring, long_ring_with_late_poles()])
    p = capi.params("launch", cluster_max_count=60)
    lim = dict(max_ring_points=4096, max_ring_candidates=2048, max_candidates=4096, max_keypoints=1024, max_total_keypoints=2048, max_kpc_points=16384)
    ora = oracle.run(p, scan)
    assert len(ora["candidates"]) > 500
    for front in (1, 0):
        fx_hooks(FX_FRONT=front)
        ctx = capi.Context(p, capi.limits(2, len(scan), **lim))
        got = ctx.process_host([scan, util.vlp16_scan(1000)[:len(scan)]])
        assert _hints(ctx)[1] == 1 and got[0]["flags"] == 0
        util.compare_scan(got[0], ora, tag=f"3000-point ring, FX_FRONT={front}")
        ctx.close()


def test_more_candidates_than_the_large_merge_tier_holds(fx_hooks, oracle):
    """The secondary merge of a scan with more candidates than the large merge tier's LDS tables hold (~16 000; the hook
    lowers it) runs on scratch in HBM — the same body, the same result."""
    fx_hooks(FX_FRONT=0, FX_MERGE_BIG_CAP=64, FX_MERGE_HUGE_CAP=128)
    p = capi.params("launch")
    scans = [util.vlp16_scan(1000 + b) for b in range(3)] + [np.concatenate([isolated_points(7 + r, 129) for r in range(4)])]
    ctx = capi.Context(p, capi.limits(4, 28800, max_ring_candidates=512, max_keypoints=512, max_total_keypoints=4096))
    got = ctx.process_host(scans)
    n_slow = _hints(ctx)[1]
    for b, s in enumerate(scans):
        util.compare_scan(got[b], oracle.run(p, s), tag=f"slow merge scan {b}")
    assert len(got[3]["candidates"]) == 516 and n_slow == 1  # (k_merge_small holds 512 candidates: the last scan goes all the way)
    ctx.close()


def test_window_boundary_points_and_duplicates(fxlib, oracle):
    """Points exactly on a ring window's edge belong to two rings (ref: node.cpp:201): the LDS ring split places them in both."""
    rng = np.random.default_rng(3)
    pts = []
    for ring in range(16):
        el_edge = -15.0 + 2.0 * ring + 1.0  # upper edge of ring `ring` = lower edge of the next
        az = np.radians(rng.uniform(-60, 60, 8))
        r = rng.uniform(8, 40, 8)
        e = np.radians(el_edge)
        pts.append(np.stack([r * np.cos(e) * np.cos(az), r * np.cos(e) * np.sin(az), r * np.sin(e), np.zeros(8)], 1))
    s = np.concatenate(pts + [util.vlp16_scan(5)[:6000, :4].astype(np.float64)]).astype(np.float32)
    got, hints = _run(oracle, [s, s[::-1].copy()], "window edges", max_keypoints=512, max_total_keypoints=1024)
    assert hints == (0, 0)


def test_many_clusters_in_one_ring_and_a_32_ring_sensor(fxlib, oracle):
    """More than 192 clusters in one ring (the order replay's partition phase by one lane), and k_front's ring capacity: 32."""
    lim = dict(max_ring_candidates=512, max_keypoints=512, max_total_keypoints=4096)
    tight = capi.params("launch", cluster_tolerance=0.5)  # (returns of one range shell 0.7 - 0.9 m apart stay clusters of their own)
    got, hints = _run(oracle, [isolated_points(9, 200), isolated_points(9, 256), np.concatenate([isolated_points(8, 190), isolated_points(9, 193)])],
                      "many clusters a ring", p=tight, **lim)
    assert hints == (0, 0) and [len(g["candidates"]) for g in got] == [200, 256, 383]
    p = capi.params("launch", n_rings=32, el0_deg=-15.0, el_step_deg=1.0, secondary_max=32, cluster_tolerance=0.5)
    scans = [isolated_points(9, 200), interleaved_arc(6, 60, step_m=0.15),
             np.concatenate([isolated_points(7, 100), long_ring_with_late_poles(), two_arcs_in_blocks(9)]), util.vlp16_scan(1000)]
    got, hints = _run(oracle, scans, "32 rings", p=p, roll=0.01, pitch=-0.01, **lim)
    assert hints == (0, 0) and [g["flags"] for g in got] == [0] * len(scans)
