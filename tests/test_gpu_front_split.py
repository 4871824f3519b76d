"""The fused front path as TWO launches (k_front_ab: streaming pass + ring split, the ring-major records through HBM;
k_front_cd: per-ring clustering + secondary merge; ref: node.cpp:147-259) — VERDICT r5 #1's split.  Built and measured in
round 6 (profiles/r06_experiments.md §1: alone the two take what the one takes, with batches in flight the one wins by 3 %),
so the product launches the one fused kernel; the test build's hook FX_FRONT_SPLIT=1 sends batches through the two launches,
FX_FRONT_SPLIT=2 through the variant whose second launch keeps only its TABLES in LDS (k_front_cdl: the ring-major points read
from HBM, 37 KB instead of 79, four 256-thread workgroups a CU — alone 0.134 ms against 0.156, in flight level with the fused
kernel: profiles/r06_experiments.md §5): the same table limits, the same hand-over to k_front_redo / k_slow, bit-identical results."""
import numpy as np
import pytest

from feature_extraction_amd import capi
from tests import util
from tests.test_gpu_front import (BIG_LIM, FRONT_CAP, FRONT_RUNS, _hints, _run, shells_ring)
from tests.test_gpu_ring_run_tier import (interleaved_arc, isolated_points, long_ring_with_late_poles, two_arcs_in_blocks)

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[1, 2], ids=["cd", "cd-lean"])
def split(request):
    """FX_FRONT_SPLIT: 1 = k_front_ab + k_front_cd (the ring-major points loaded into k_front's image), 2 = k_front_ab +
    k_front_cdl (the lean image: tables only, the points read from B.ring_pts; 256 threads, four workgroups a CU)."""
    return request.param


def test_a_64_scan_batch_through_the_two_launches_equals_the_fused_kernel(fxlib, fx_hooks, oracle, split):
    """64 bench scans (with an empty and a ragged one): the two launches against the oracle, and bit-identical to the product's
    one fused launch."""
    scans = [util.vlp16_scan(3000 + b) for b in range(64)]
    scans[5] = np.zeros((0, 4), np.float32)
    scans[9] = scans[9][:7000]
    p = capi.params("launch")
    lim = capi.limits(64, 28800)
    keys = ("filtered", "candidates", "cand_size", "kpc", "kpc_cand", "cand_keypoint", "keypoints", "kp_size", "kp_neighbors", "descriptors")
    res = {}
    for mode in (0, split):
        fx_hooks(FX_FRONT_SPLIT=mode)
        ctx = capi.Context(p, lim)
        res[min(mode, 1)] = ctx.process_host(scans, roll=0.02, pitch=-0.015)
        assert _hints(ctx) == (0, 0)
        ctx.close()
    for b in (0, 5, 9, 33, 63):
        util.compare_scan(res[1][b], oracle.run(p, scans[b], roll=0.02, pitch=-0.015), tag=f"64-scan batch, two launches ({split}), scan {b}")
    for b in range(64):
        for key in keys:
            util.assert_bit_equal(res[0][b][key], res[1][b][key], f"fused vs two launches, scan {b} {key}")


def test_ring_shapes_and_bench_scans(fx_hooks, oracle, split):
    fx_hooks(FX_FRONT_SPLIT=split)
    scans = [long_ring_with_late_poles(), two_arcs_in_blocks(3), interleaved_arc(12, 60), isolated_points(14, 100), isolated_points(9, 129),
             interleaved_arc(6, 100),
             np.concatenate([long_ring_with_late_poles(), two_arcs_in_blocks(9), interleaved_arc(6, 36), isolated_points(10, 128)]),
             np.zeros((0, 4), np.float32)]
    got, hints = _run(oracle, scans, "split shapes", max_ring_candidates=512, max_keypoints=512, max_total_keypoints=4096)
    assert [g["flags"] for g in got] == [0] * len(scans) and hints == (0, 0)
    scans = [util.vlp16_scan(1000 + b) for b in range(12)]
    for preset in ("launch", "default"):
        got, hints = _run(oracle, scans, f"split {preset}", p=capi.params(preset), roll=0.02, pitch=-0.015)
        assert hints == (0, 0) and sum(g["n_keypoints"] for g in got) > 0


def test_table_limits_hand_over_to_the_kernels_behind(fx_hooks, oracle, split):
    """k_front_cd's run table / near-pair list at their limits; more ring entries than either launch holds (k_front_ab hands the
    scan on itself)."""
    fx_hooks(FX_FRONT_SPLIT=split)

    def runs(total):
        per, rest = divmod(total, 4)
        return np.concatenate([isolated_points(7 + r, per + (1 if r < rest else 0)) for r in range(4)])
    lim = dict(max_ring_candidates=512, max_keypoints=512, max_total_keypoints=4096)
    got, hints = _run(oracle, [runs(FRONT_RUNS)], "split: runs at the limit", **lim)
    assert hints == (0, 0) and got[0]["flags"] == 0 and len(got[0]["candidates"]) == FRONT_RUNS
    got, hints = _run(oracle, [runs(FRONT_RUNS + 1), runs(FRONT_RUNS)], "split: runs past the limit", **lim)
    assert hints == (1, 0) and [g["flags"] for g in got] == [0, 0]
    three = np.concatenate([interleaved_arc(6, 100), interleaved_arc(7, 100), interleaved_arc(8, 100)])
    got, hints = _run(oracle, [interleaved_arc(6, 100), three], "split: near pairs")
    assert hints == (1, 0) and [g["flags"] for g in got] == [0, 0]
    p = capi.params("launch", x_min=-100.0)
    scans = [util.vlp16_scan(1000 + b) for b in range(4)]
    got, hints = _run(oracle, scans, "split: full circle", p=p, roll=0.02, pitch=-0.015)
    assert all(len(g["filtered"]) > FRONT_CAP and g["flags"] == 0 for g in got) and hints == (4, 0)


@pytest.mark.parametrize("force", [1, 2])
def test_every_scan_through_the_kernels_behind(fx_hooks, oracle, force, split):
    fx_hooks(FX_FRONT_SPLIT=split, FX_FRONT_FORCE=force)
    scans = [util.vlp16_scan(1000 + b) for b in range(6)] + [np.zeros((0, 4), np.float32), long_ring_with_late_poles(), interleaved_arc(6, 100)]
    got, hints = _run(oracle, scans, f"split forced {force}", roll=0.02, pitch=-0.015)
    assert [g["flags"] for g in got] == [0] * len(scans)
    assert hints == (len(scans) - 1, len(scans) - 1 if force == 2 else 0)


def test_interleaving_and_the_slow_tier(fx_hooks, oracle, split):
    """Per-scan results are a function of the scan alone through the two launches too; a ring beyond the LDS tiers is exact on
    its first presentation."""
    fx_hooks(FX_FRONT_SPLIT=split)
    p = capi.params("launch")
    big = np.concatenate([shells_ring(), long_ring_with_late_poles()])
    pool = {"small0": util.vlp16_scan(1000), "big": big, "runs": np.concatenate([isolated_points(7 + r, 129) for r in range(4)]),
            "empty": np.zeros((0, 4), np.float32)}
    lim = dict(BIG_LIM, max_total_keypoints=4096)
    ctx = capi.Context(p, capi.limits(4, 28800, **lim))
    ref = {}
    for name, s in pool.items():
        ctx.process_host([pool["small0"]])
        ref[name] = ctx.process_host([s])[0]
        assert ref[name]["flags"] == 0, name
    util.compare_scan(ref["big"], oracle.run(p, big), tag="split: big alone")
    util.compare_scan(ref["runs"], oracle.run(p, pool["runs"]), tag="split: runs alone")
    rng = np.random.default_rng(6)
    names = list(pool)
    for trial in range(8):
        batch = [names[i] for i in rng.integers(0, len(names), int(rng.integers(1, 5)))]
        got = ctx.process_host([pool[n] for n in batch])
        for n, g in zip(batch, got):
            for key in ("filtered", "candidates", "cand_size", "kpc", "kpc_cand", "cand_keypoint", "keypoints", "kp_size", "kp_neighbors", "descriptors"):
                util.assert_bit_equal(g[key], ref[n][key], f"split trial {trial} {batch} {n} {key}")
    ctx.close()


def test_window_boundary_points_and_a_32_ring_sensor(fx_hooks, oracle, split):
    fx_hooks(FX_FRONT_SPLIT=split)
    rng = np.random.default_rng(3)
    pts = []
    for ring in range(16):
        e = np.radians(-15.0 + 2.0 * ring + 1.0)
        az = np.radians(rng.uniform(-60, 60, 8))
        r = rng.uniform(8, 40, 8)
        pts.append(np.stack([r * np.cos(e) * np.cos(az), r * np.cos(e) * np.sin(az), r * np.sin(e), np.zeros(8)], 1))
    s = np.concatenate(pts + [util.vlp16_scan(5)[:6000, :4].astype(np.float64)]).astype(np.float32)
    got, hints = _run(oracle, [s, s[::-1].copy()], "split: window edges", max_keypoints=512, max_total_keypoints=1024)
    assert hints == (0, 0)
    lim = dict(max_ring_candidates=512, max_keypoints=512, max_total_keypoints=4096)
    p = capi.params("launch", n_rings=32, el0_deg=-15.0, el_step_deg=1.0, secondary_max=32, cluster_tolerance=0.5)
    scans = [isolated_points(9, 200), interleaved_arc(6, 60, step_m=0.15),
             np.concatenate([isolated_points(7, 100), long_ring_with_late_poles(), two_arcs_in_blocks(9)]), util.vlp16_scan(1000)]
    got, hints = _run(oracle, scans, "split: 32 rings", p=p, roll=0.01, pitch=-0.01, **lim)
    assert hints == (0, 0) and [g["flags"] for g in got] == [0] * len(scans)
