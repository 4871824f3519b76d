"""The fused front kernel (k_front: filter, ring split, per-ring clustering and secondary merge of a scan in ONE launch, the
intermediates in LDS; ref: node.cpp:147-259) and the two kernels behind it: k_front_redo (the general kernels' bodies in
k_front's shape, for scans that do not fit its tables) and k_tail (the same in a whole CU, for what does not fit that
either).  Which of them took a scan is read from the context's tier hints (6: scans handed to k_front_redo, 7: to k_tail);
every result is compared with the oracle."""
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
    """The test build's hook hands every scan to k_front_redo (1), and from there to k_tail (2)."""
    fx_hooks(FX_FRONT_FORCE=force)
    scans = [util.vlp16_scan(1000 + b) for b in range(6)] + [np.zeros((0, 4), np.float32), long_ring_with_late_poles(), interleaved_arc(6, 100)]
    for preset in ("launch", "default"):
        got, hints = _run(oracle, scans, f"forced {force} {preset}", p=capi.params(preset), roll=0.02, pitch=-0.015)
        assert [g["flags"] for g in got] == [0] * len(scans)
        assert hints == (len(scans) - 1, len(scans) - 1 if force == 2 else 0)  # (the empty scan never gets that far)


def test_a_scan_that_needs_the_whole_cu_kernel(fxlib, oracle):
    """A ring of 1700 points in 420 clusters (six range shells, visited in azimuth order) fits neither k_front's tables nor
    k_front_redo's LDS image: k_tail's work.  A fresh context launches k_tail (nothing known yet); a context whose previous
    batch needed nothing behind k_front does not — the scan then comes back FLAGGED (never silently wrong), and the next
    batch, sized by what that one needed, is exact."""
    az, rg = [], []
    for shell in (20.0, 25.0, 30.0, 35.0, 40.0, 45.0):
        step = np.degrees(1.45 / shell)  # cluster centres 1.45 m apart along the shell (the tolerance is 1 m)
        for c in np.arange(-86.0, 86.0, step):
            az += list(c + 0.1 * np.arange(4))
            rg += [shell] * 4
    order = np.argsort(np.array(az), kind="stable")
    shells = ring_points(9, np.array(az)[order], np.array(rg)[order])
    assert len(shells) > 1600
    big = np.concatenate([shells, long_ring_with_late_poles()])
    small = util.vlp16_scan(1000)
    p = capi.params("launch")
    lim = dict(max_ring_candidates=2048, max_candidates=4096, max_keypoints=512, max_total_keypoints=1024, max_kpc_points=8192)
    got, hints = _run(oracle, [big], "whole-CU tier, fresh context", p=p, **lim)
    assert hints == (1, 1) and got[0]["flags"] == 0 and len(got[0]["candidates"]) > 400
    ctx = capi.Context(p, capi.limits(1, 28800, **lim))
    _, hints = _run(oracle, [small], "whole-CU tier, before", p=p, ctx=ctx)
    assert hints == (0, 0)
    got = ctx.process_host([big])
    assert _hints(ctx) == (1, 1)
    assert got[0]["flags"] & (capi.FX_FLAG_RING_OVERFLOW | capi.FX_FLAG_CAND_OVERFLOW) and got[0]["n_keypoints"] == 0
    got, hints = _run(oracle, [big], "whole-CU tier, the batch after", p=p, ctx=ctx)
    assert hints == (1, 1) and got[0]["flags"] == 0
    _, hints = _run(oracle, [small], "whole-CU tier, small again", p=p, ctx=ctx)
    assert hints == (0, 0)
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
