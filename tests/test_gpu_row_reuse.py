"""Descriptor rows outlive a batch: k_desc_group clears a row by un-writing the bins it wrote to it last time (rows another
tier or a NaN fill wrote are cleared whole).  One context, different batches one after another — every row must come out as
if it had been cleared in full: compared with the oracle, and with a fresh context, bit for bit."""
import numpy as np
import pytest

from feature_extraction_amd import capi
from tests import util

pytestmark = pytest.mark.gpu


def _check(ctx, oracle, p, scans, tag, roll=0.02, pitch=-0.015):
    got = ctx.process_host(scans, roll=roll, pitch=pitch)
    for b, s in enumerate(scans):
        util.compare_scan(got[b], oracle.run(p, s, roll=roll, pitch=pitch), tag=f"{tag} scan {b}")
    return got


def test_rows_are_clean_across_batches_of_different_scenes(fxlib, oracle):
    p = capi.params("launch")
    ctx = capi.Context(p, capi.limits(6, 28800, max_total_keypoints=6 * 256))
    sparse = [util.vlp16_scan(3000 + b) for b in range(6)]
    dense = [util.vlp16_scan(3100 + b, n_poles=256) for b in range(4)]      # more rows, more bins per row
    lonely = [util.vlp16_scan(3200 + b, n_poles=8) for b in range(3)]       # few rows
    _check(ctx, oracle, p, dense, "dense first")
    _check(ctx, oracle, p, sparse, "sparse after dense")      # rows that held many bins now hold few
    _check(ctx, oracle, p, lonely, "few rows after many")     # most rows of the last batch are not touched at all ...
    _check(ctx, oracle, p, dense[::-1], "dense again")        # ... and are reused here with their old lists
    _check(ctx, oracle, p, [sparse[0]], "one scan")
    ctx.close()


def test_rows_written_by_the_other_tiers_and_nan_rows_are_cleared_whole(fxlib, oracle):
    # descriptor radius 1.2 m: most keypoints leave the 64-point tier (wavefront and list tiers write their rows); support
    # radius 0.05 m: most keypoints have no neighbour at all (NaN rows); then the launch preset on the same rows
    base = capi.params("launch")
    big = capi.params("launch", descriptor_radius=1.2)
    tiny = capi.params("launch", descriptor_radius=0.05)
    scans = [util.vlp16_scan(3300 + b, n_poles=128) for b in range(3)]
    lim = capi.limits(3, 28800, max_total_keypoints=3 * 256)
    # (a context has one parameter set: the rows' history is made with three contexts' worth of batches on one device
    #  buffer only within a context, so each parameter set gets its own context and runs dense -> plain -> dense)
    for p, tag in ((big, "R 1.2 m"), (tiny, "R 0.05 m"), (base, "launch")):
        ctx = capi.Context(p, lim)
        first = _check(ctx, oracle, p, scans, tag + " a")
        _check(ctx, oracle, p, scans[::-1], tag + " b")
        again = _check(ctx, oracle, p, scans, tag + " c")
        for a, b in zip(first, again):
            assert np.array_equal(a["descriptors"].view(np.uint32), b["descriptors"].view(np.uint32))
        ctx.close()


def test_row_history_with_a_changing_batch_size_matches_a_fresh_context(fxlib):
    p = capi.params("launch")
    lim = capi.limits(8, 28800, max_total_keypoints=8 * 256)
    ctx = capi.Context(p, lim)
    rng = np.random.default_rng(5)
    pool = [util.vlp16_scan(3400 + b, n_poles=int(rng.integers(8, 256))) for b in range(10)]
    for step in range(12):
        pick = [pool[i] for i in rng.choice(len(pool), size=int(rng.integers(1, 9)), replace=False)]
        got = ctx.process_host(pick, roll=0.01, pitch=0.0)
        fresh_ctx = capi.Context(p, lim)
        want = fresh_ctx.process_host(pick, roll=0.01, pitch=0.0)
        fresh_ctx.close()
        for b, (g, w) in enumerate(zip(got, want)):
            assert g["flags"] == 0 and w["flags"] == 0
            assert g["n_keypoints"] == w["n_keypoints"]
            assert np.array_equal(g["descriptors"].view(np.uint32), w["descriptors"].view(np.uint32)), f"step {step} scan {b}"
    ctx.close()


def test_tier_grids_follow_the_previous_batch_without_changing_results(fx_hooks, oracle):
    """The grids of the rarely used tiers follow the context's previous batch (FxBuffers::tier_hint): a dense batch right
    after sparse ones runs its dense tier on the smallest grids, a sparse one after it on wide ones — same results either
    way, and with FX_TIER_MIN_GRID=0 (always the full grids)."""
    import ctypes as C
    p = capi.params("default", descriptor_radius=4.0)
    dense = util.vlp16_scan(1000, n_poles=8, x_lo=3.0, x_hi=8.0, y_lo=-4.0, y_hi=4.0)  # rows of > 1024 support points
    sparse = util.vlp16_scan(1001, n_poles=4)
    ora_d, ora_s = oracle.run(p, dense), oracle.run(p, sparse)
    assert ora_d["kp_neighbors"].max() > 1100
    lib = capi.load()
    lib.fx_debug_tier_hints.argtypes = [C.c_void_p, C.c_void_p]
    for min_grid in ("8", "1", "0"):
        fx_hooks(FX_TIER_MIN_GRID=min_grid)
        ctx = capi.Context(p, capi.limits(2, 28800))
        seen = []
        for scans, oras, tag in (([sparse], [ora_s], "sparse"), ([sparse, sparse], [ora_s, ora_s], "sparse x2"),
                                 ([dense, sparse], [ora_d, ora_s], "dense after sparse"), ([dense, dense], [ora_d, ora_d], "dense x2"),
                                 ([sparse], [ora_s], "sparse after dense"), ([dense], [ora_d], "dense again")):
            got = ctx.process_host(scans)
            for b, o in enumerate(oras):
                util.compare_scan(got[b], o, tag=f"min grid {min_grid}: {tag} scan {b}")
            hints = (C.c_uint32 * 8)()
            capi.check(lib.fx_debug_tier_hints(ctx.handle, hints))
            seen.append(int(hints[4]))
        ctx.close()
        assert seen[0] == 0 and seen[2] > 0 and seen[3] > seen[2] and seen[4] == 0, seen  # dense rows of each batch


def test_a_failed_batch_leaves_nothing_for_the_next_one_to_build_on(fx_hooks, oracle):
    """a batch that fails after its kernels were enqueued (the test build's hook returns FX_ERR_HIP from the third
    fx_process_batch at that point) — and whatever scribbled into the descriptor rows meanwhile — must not show in the next batch:
    every row is then cleared whole, counters and tier hints start over.  Compared with the oracle and with a fresh context."""
    import torch
    fx_hooks(FX_FAIL_AFTER_ENQUEUE=3)
    p = capi.params("launch")
    lim = capi.limits(4, 28800, max_total_keypoints=4 * 256)
    ctx = capi.Context(p, lim)
    a = [util.vlp16_scan(3400 + b, n_poles=128) for b in range(4)]
    b = [util.vlp16_scan(3500 + b) for b in range(3)]
    _check(ctx, oracle, p, a, "before the failure, a")
    _check(ctx, oracle, p, b, "before the failure, b")
    with pytest.raises(capi.FxError, match="FX_FAIL_AFTER_ENQUEUE"):
        ctx.process_host(a)
    ctx.synchronize()
    # what a half-finished pipeline (or a caller that ignored the READ-ONLY rule of d_descriptors) might leave in the rows
    descs = ctx.make_descs([0], [0])
    v = ctx.process_raw(descs, 0, 0)  # (an empty batch: the view's device pointers)
    import ctypes as C
    n_f4 = 4 * 256 * capi.FX_DESC_FLOATS // 4  # the whole pool, as 16-byte records: written through the library's own unpack kernel
    junk = torch.full((n_f4, 4), 7.0, dtype=torch.float32, device="cuda")
    lay = capi.FxPc2Layout(16, 0, 4, 8, 12, 0)
    capi.check(ctx.lib.fx_unpack_pointcloud2(ctx.handle, C.c_void_p(junk.data_ptr()), n_f4, C.byref(lay), C.c_void_p(v.d_descriptors)))
    ctx.synchronize()
    fx_hooks(FX_FAIL_AFTER_ENQUEUE=0)
    got = _check(ctx, oracle, p, b[::-1], "after the failure")
    fresh = capi.Context(p, lim)
    want = fresh.process_host(b[::-1], roll=0.02, pitch=-0.015)
    for g, w in zip(got, want):
        assert np.array_equal(g["descriptors"].view(np.uint32), w["descriptors"].view(np.uint32))
    _check(ctx, oracle, p, a, "and the batch after that")
    fresh.close()
    ctx.close()
