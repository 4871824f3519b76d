"""Edge cases of the hot path on the GPU: empty / tiny / ragged scans, non-finite points,
elevation-window boundaries, unordered input, strides, device-resident input, determinism,
capacity flags."""
import ctypes as C

import numpy as np
import pytest

from feature_extraction_amd import capi
from tests import util

pytestmark = pytest.mark.gpu


def _cmp(oracle, p, lim, scans, roll=0.0, pitch=0.0, tag=""):
    ctx = capi.Context(p, lim)
    got = ctx.process_host(scans, roll=roll, pitch=pitch)
    for b, s in enumerate(scans):
        ora = oracle.run(p, s, roll=roll, pitch=pitch)
        util.compare_scan(got[b], ora, tag=f"{tag} scan {b}")
    ctx.close()
    return got


def test_empty_and_ragged_batch(fxlib, oracle):
    """Empty scans give K = 0 (ref: node.cpp:209-210, 263-264, 331-332 early returns)."""
    full = util.vlp16_scan(1000)
    scans = [np.zeros((0, 4), np.float32), full[:1], full[:17], full[:16 * 100], full, np.zeros((0, 4), np.float32),
             full[5000:9000]]
    got = _cmp(oracle, capi.params("launch"), capi.limits(len(scans), 28800), scans, 0.02, -0.015, "ragged")
    assert got[0]["n_keypoints"] == 0 and len(got[0]["filtered"]) == 0 and got[4]["n_keypoints"] > 0


def test_host_results_of_a_stream_of_calls_with_changing_keypoint_counts(fxlib, oracle):
    """FX_OUT_HOST: the per-scan words (counts, offsets, flags) come back in one copy of one block, the descriptor rows in a copy
    sized by the batch's own total.  One context, a scan a call: sparse, crowded, empty, sparse again, then a batch of three —
    every call's results equal the oracle's, whatever the call before left in the host buffers."""
    p = capi.params("launch")
    sparse, crowded = util.vlp16_scan(1000, n_poles=6), util.vlp16_scan(7, n_poles=400)
    ora = {id(s): oracle.run(p, s, roll=0.02, pitch=-0.015) for s in (sparse, crowded)}
    assert ora[id(crowded)]["n_keypoints"] > 4 * max(1, ora[id(sparse)]["n_keypoints"])
    ctx = capi.Context(p, capi.limits(3, 28800, max_keypoints=512, max_total_keypoints=1536, max_kpc_points=8192, max_candidates=3500))
    empty = np.zeros((0, 4), np.float32)
    for step, scans in enumerate(([sparse], [crowded], [empty], [sparse], [crowded, sparse, crowded], [sparse])):
        got = ctx.process_host(scans, roll=0.02, pitch=-0.015)
        for b, sc in enumerate(scans):
            if len(sc):
                util.compare_scan(got[b], ora[id(sc)], tag=f"call {step} scan {b}")
            else:
                assert got[b]["n_keypoints"] == 0 and got[b]["flags"] == 0
    ctx.close()


def test_empty_batch_call(fxlib):
    ctx = capi.Context(capi.params("default"), capi.limits(4, 1024))
    v = ctx.process_raw(ctx.make_descs([], []), 0, capi.FX_OUT_HOST)
    assert v.batch == 0 and v.total_keypoints == 0
    ctx.close()


def test_everything_filtered_out(fxlib, oracle):
    s = util.vlp16_scan(3)
    s[:, 0] = -np.abs(s[:, 0]) - 1.0  # x_min = 0 removes every point
    got = _cmp(oracle, capi.params("default"), capi.limits(1, 28800), [s], tag="all filtered")
    assert got[0]["n_keypoints"] == 0 and len(got[0]["filtered"]) == 0


def test_non_finite_points_are_dropped(fxlib, oracle):
    """PassThrough removes non-finite points (A.3); they are not part of the 3DSC search surface."""
    s = util.vlp16_scan(1000)
    rng = np.random.default_rng(0)
    bad = rng.choice(len(s), 600, replace=False)
    s[bad[:200], 0] = np.nan
    s[bad[200:400], 1] = np.inf
    s[bad[400:], 2] = -np.inf
    _cmp(oracle, capi.params("launch"), capi.limits(1, 28800), [s], 0.02, -0.015, "NaN/Inf injection")


def test_points_on_window_boundaries_belong_to_two_rings(fxlib, oracle):
    """Inclusive ring windows (ref: node.cpp:201): an even-degree elevation sits in two rings."""
    n = 48  # 8 groups of 6 points, groups 1.2 m apart, points 3 cm apart
    rho = 8.0
    pts = np.zeros((4 * n, 4), np.float32)
    for i, el_deg in enumerate((0.0, 2.0, -4.0, 1.0)):
        el = np.radians(el_deg)
        a = 0.1 + 0.15 * (np.arange(n) // 6) + (np.arange(n) % 6) * 0.004
        pts[i * n:(i + 1) * n, 0] = rho * np.cos(a)
        pts[i * n:(i + 1) * n, 1] = rho * np.sin(a)
        pts[i * n:(i + 1) * n, 2] = rho * np.tan(el)
    p = capi.params("default", z_min=-3.0)
    got = _cmp(oracle, p, capi.limits(1, 4 * n), [pts], tag="boundary elevations")
    ora = oracle.run(p, pts, want_labels=True)
    assert ((ora["ring_labels"] >= 0).sum(axis=0) == 2).sum() >= 2 * n  # points really are in two rings
    assert got[0]["n_keypoints"] > 0


def test_boundary_points_among_many_rings(fxlib, oracle):
    """A 64-ring sensor in firing order has every ring in every 512-point chunk: k_bucket then ranks a wavefront's points by
    sorting (ring, lane) keys — except in wavefronts that hold a point on a window boundary (two rings), which take the
    per-ring ballots; both kinds share the chunk's per-ring offsets."""
    over = dict(n_rings=64, n_az=256, el0_deg=-31.5, el_step_deg=1.0)
    s = util.vlp16_scan(4100, **over)
    s[::53, 2] = 0.0  # elevation exactly 0: the boundary of rings 31 and 32 (windows [-1, 0] and [0, 1])
    p = capi.params("default", z_min=-3.0, secondary_max=64, **{k: over[k] for k in ("n_rings", "el0_deg", "el_step_deg")})
    lim = capi.limits(1, len(s), max_candidates=4096, max_kpc_points=32768, max_keypoints=512, max_total_keypoints=512)
    got = _cmp(oracle, p, lim, [s], tag="64 rings, boundary points")
    ora = oracle.run(p, s, want_labels=True)
    assert ((ora["ring_labels"] >= 0).sum(axis=0) == 2).sum() > 50  # points really are in two rings
    assert len(got[0]["filtered"]) > 4000


def test_unordered_input_takes_the_all_pairs_path(fxlib, oracle):
    """Shuffled points: no azimuth order, one run per point — exactness must not depend on order."""
    rng = np.random.default_rng(4)
    s = util.vlp16_scan(1000)
    s = s[rng.permutation(len(s))]
    _cmp(oracle, capi.params("launch"), capi.limits(1, 28800), [s], 0.02, -0.015, "shuffled")
    _cmp(oracle, capi.params("default"), capi.limits(1, 28800), [s], 0.02, -0.015, "shuffled default")


def test_pcl_stride32_and_device_input(fxlib, oracle):
    import torch
    p = capi.params("launch")
    s = util.vlp16_scan(1000)
    ora = oracle.run(p, s, roll=0.02, pitch=-0.015)
    ctx = capi.Context(p, capi.limits(2, 28800))
    # pcl::PointXYZI in-memory layout: x y z pad intensity pad pad pad (32 bytes)
    wide = np.zeros((len(s), 8), np.float32)
    wide[:, :3] = s[:, :3]
    wide[:, 3] = 1.0
    wide[:, 4] = 123.0
    got = ctx.process_host([wide], roll=0.02, pitch=-0.015)[0]
    util.compare_scan(got, ora, tag="host stride 32")
    for arr, stride in ((s, 16), (wide, 32)):
        d = torch.from_numpy(arr).cuda()
        descs = ctx.make_descs([d.data_ptr()], [len(arr)], stride, 0.02, -0.015)
        flags = capi.FX_IN_DEVICE | capi.FX_OUT_HOST | capi.FX_OUT_CLOUDS | capi.FX_OUT_DEBUG
        got = ctx.unpack(ctx.process_raw(descs, 1, flags))[0]
        util.compare_scan(got, ora, tag=f"device stride {stride}")
    ctx.close()


def test_host_batches_of_mixed_record_strides(fxlib, oracle):
    """Host scans go to the device as they are: ragged batches that mix 16-byte and 32-byte (pcl::PointXYZI) records, a
    context that first sees 16-byte records and then wider ones (the staging buffer grows), an empty scan in between."""
    p = capi.params("launch")
    base = [util.vlp16_scan(1000 + b) for b in range(5)]
    base[1] = base[1][:17000]
    base[3] = base[3][:9001]
    oras = [oracle.run(p, s, roll=0.02, pitch=-0.015) for s in base]
    def wide(s):
        w = np.full((len(s), 8), 7.0, np.float32)
        w[:, :3] = s[:, :3]
        return w
    ctx = capi.Context(p, capi.limits(6, 28800))
    flags = capi.FX_OUT_HOST | capi.FX_OUT_CLOUDS | capi.FX_OUT_DEBUG
    for strides in ([16] * 5, [32, 16, 32, 32, 16], [32] * 5, [16, 32, 16, 16, 32]):
        arrs = [wide(s) if st == 32 else s for s, st in zip(base, strides)]
        arrs.insert(2, np.zeros((0, 4), np.float32))
        st_all = strides[:2] + [16] + strides[2:]
        descs = ctx.make_descs([a.ctypes.data if len(a) else 0 for a in arrs], [len(a) for a in arrs], 16, 0.02, -0.015)
        for d, st in zip(descs, st_all):
            d.stride_bytes = st
        got = ctx.unpack(ctx.process_raw(descs, len(arrs), flags))
        assert got[2]["n_keypoints"] == 0 and len(got[2]["filtered"]) == 0
        for b, (g, o) in enumerate(zip(got[:2] + got[3:], oras)):
            util.compare_scan(g, o, tag=f"strides {strides} scan {b}")
    ctx.close()


def test_argument_errors(fxlib):
    ctx = capi.Context(capi.params("default"), capi.limits(2, 1000))
    s = util.vlp16_scan(1)[:500]
    v = capi.FxBatchView()
    assert fxlib.fx_process_batch(ctx.handle, ctx.make_descs([s.ctypes.data] * 3, [500] * 3), 3, 0, C.byref(v)) == 5
    assert fxlib.fx_process_batch(ctx.handle, ctx.make_descs([s.ctypes.data], [5000]), 1, 0, C.byref(v)) == 5
    assert fxlib.fx_process_batch(ctx.handle, ctx.make_descs([s.ctypes.data], [500], stride_bytes=12), 1, 0, C.byref(v)) == 1
    assert fxlib.fx_process_batch(ctx.handle, ctx.make_descs([s.ctypes.data + 4], [400]), 1, 0, C.byref(v)) == 1
    ctx.close()
    with pytest.raises(capi.FxError):
        capi.Context(capi.params("default", n_rings=0), capi.limits(2, 1000))
    with pytest.raises(capi.FxError):
        capi.Context(capi.params("default"), capi.limits(2, 1000, max_ring_points=100000))


def test_determinism_and_context_reuse(fxlib):
    """Same input twice, and through a second context: bit-identical (no float atomics anywhere)."""
    scans = [util.vlp16_scan(1000 + b) for b in range(4)]
    p, lim = capi.params("launch"), capi.limits(4, 28800)
    ctx = capi.Context(p, lim)
    a = ctx.process_host(scans, roll=0.02, pitch=-0.015)
    ctx.process_host(scans[::-1], roll=0.0, pitch=0.0)  # different batch in between
    b = ctx.process_host(scans, roll=0.02, pitch=-0.015)
    ctx2 = capi.Context(p, lim)
    c = ctx2.process_host(scans, roll=0.02, pitch=-0.015)
    for x, y, z in zip(a, b, c):
        for k in ("filtered", "candidates", "keypoints", "descriptors", "kpc", "cand_keypoint", "kp_neighbors"):
            util.assert_bit_equal(x[k], y[k], k)
            util.assert_bit_equal(x[k], z[k], k)
    ctx.close()
    ctx2.close()


def test_streaming_batch_of_one(fxlib, oracle):
    """B = 1 at sensor rate (the ROS shell's mode) through one long-lived context."""
    p = capi.params("launch")
    ctx = capi.Context(p, capi.limits(1, 28800))
    for seed in (1, 2, 3):
        s = util.vlp16_scan(seed)
        got = ctx.process_host([s], roll=0.01 * seed, pitch=-0.01)[0]
        util.compare_scan(got, oracle.run(p, s, roll=0.01 * seed, pitch=-0.01), tag=f"stream {seed}")
    ctx.close()


def test_capacity_overflow_is_flagged_never_silent(fxlib, oracle):
    s = util.vlp16_scan(1000)
    p = capi.params("launch")
    ora = oracle.run(p, s, roll=0.02, pitch=-0.015)
    for over, bit in ((dict(max_keypoints=8), 0x4), (dict(max_ring_candidates=4), 0x2), (dict(max_candidates=64), 0x2),
                      (dict(max_ring_points=128), 0x1), (dict(max_total_keypoints=10), 0x10),
                      (dict(max_kpc_points=100), 0x20)):
        ctx = capi.Context(p, capi.limits(1, 28800, **over))
        got = ctx.process_host([s], roll=0.02, pitch=-0.015)[0]
        assert got["flags"] & bit, (over, hex(got["flags"]))
        ctx.close()
    # and with room for everything the flags stay clear
    ctx = capi.Context(p, capi.limits(1, 28800))
    got = ctx.process_host([s], roll=0.02, pitch=-0.015)[0]
    assert got["flags"] == 0 and got["n_keypoints"] == ora["n_keypoints"]
    ctx.close()


def test_overflowed_lists_take_the_dense_tier(fxlib, oracle):
    """max_neighbors only sizes the per-row lists: what does not fit goes to the scan's overflow region and the row to
    the dense tier (cell-sorted pools, per-scan density cache), whose results are the same."""
    s = util.vlp16_scan(1000)
    p = capi.params("launch")
    _cmp(oracle, p, capi.limits(1, 28800, max_neighbors=64), [s], 0.02, -0.015, "dense tier (64-entry lists)")
    _cmp(oracle, p, capi.limits(1, 28800, max_neighbors=16), [s], 0.02, -0.015, "dense tier (16-entry lists)")
    _cmp(oracle, capi.params("default"), capi.limits(1, 28800, max_neighbors=128), [s], 0.02, -0.015, "dense tier, default preset")
    # several scans share nothing: each has its own overflow region and density cache
    scans = [util.vlp16_scan(1000 + b) for b in range(3)]
    _cmp(oracle, p, capi.limits(3, 28800, max_neighbors=32), scans, 0.02, -0.015, "dense tier, batch of 3")


def test_dense_tier_key_sort_in_global_memory(fx_hooks, oracle):
    """k_dense_finish sorts up to 14336 keys in LDS; rows beyond that sort in their region of the key pool. Reached
    here by lowering the LDS capacity (FX_DENSE_LDS_KEYS is read at fx_create; it can only lower it)."""
    s = util.vlp16_scan(1000)
    fx_hooks(FX_DENSE_LDS_KEYS=20)
    _cmp(oracle, capi.params("launch"), capi.limits(1, 28800, max_neighbors=64), [s], 0.02, -0.015, "dense tier, global key sort")


def test_dense_tier_query_marks_in_the_sorted_region(fx_hooks, oracle):
    """k_dense_sort remembers which support points a row won (whose density it computes) in an LDS bit map of 65536 sorted
    positions; rows with more support points mark them in the sorted region itself. Reached here by lowering that capacity
    (FX_DENSE_WON_POINTS, read at fx_create)."""
    fx_hooks(FX_DENSE_WON_POINTS=16)
    s = util.vlp16_scan(1000)
    _cmp(oracle, capi.params("launch"), capi.limits(1, 28800, max_neighbors=64), [s], 0.02, -0.015, "dense tier, marks in the sorted region")
    s = util.vlp16_scan(1000, n_poles=8, x_lo=3.0, x_hi=8.0, y_lo=-4.0, y_hi=4.0)
    _cmp(oracle, capi.params("default", descriptor_radius=4.0), capi.limits(1, 28800), [s], tag="dense neighbourhoods, marks in the sorted region")


@pytest.mark.parametrize("dense_slow", [0, 1])
def test_dense_tier_pool_exhaustion_is_flagged(fx_hooks, oracle, dense_slow):
    """A sorted pool too small for the batch's dense rows: FX_FLAG_NBR_OVERFLOW on the scan, NaN descriptors for the rows
    that did not fit, every other row still exact — whether the tier's own kernels run or the one small launch that stands
    in for them (the pool accounting is k_desc_group's, before either)."""
    fx_hooks(FX_DENSE_SLOW=dense_slow)
    s = util.vlp16_scan(1000, n_poles=8, x_lo=3.0, x_hi=8.0, y_lo=-4.0, y_hi=4.0)
    p = capi.params("default", descriptor_radius=4.0)
    ora = oracle.run(p, s)
    assert ora["kp_neighbors"].max() > 1100
    ctx = capi.Context(p, capi.limits(1, 28800, max_dense_points=4096))
    got = ctx.process_host([s])[0]
    ctx.close()
    assert got["flags"] & 0x8
    K = ora["n_keypoints"]
    assert got["n_keypoints"] == K
    bad = [k for k in range(K) if np.isnan(got["descriptors"][k, :1980]).all() and not np.isnan(ora["descriptors"][k, :1980]).all()]
    assert bad, "a 4096-entry pool cannot hold this scan's dense rows"
    for k in range(K):
        if k not in bad:
            np.testing.assert_allclose(got["descriptors"][k], ora["descriptors"][k], rtol=0, atol=util.DESC_TOL)


def test_sparse_limits_preset_gives_the_same_results_and_flags_what_it_cannot_hold(fxlib, oracle):
    """fx_limits_sparse (new in 0.6: small dense-tier pools and overflow regions, for VLP-16-class workloads — 4.5 GB a
    1024-scan context instead of 7): the bench scans and a scan with a few dense rows come back exactly as with the default
    limits; a scan whose rows overflow their lists by more than the preset's regions hold is FLAGGED, never silently short."""
    p = capi.params("launch")
    scans = [util.vlp16_scan(1000 + b) for b in range(3)]
    _cmp(oracle, p, capi.limits(3, 28800, sparse=True), scans, 0.02, -0.015, "sparse preset, bench scans")
    dense = util.vlp16_scan(1000, n_poles=8, x_lo=3.0, x_hi=8.0, y_lo=-4.0, y_hi=4.0)
    pd = capi.params("default", descriptor_radius=4.0)
    assert oracle.run(pd, dense)["kp_neighbors"].max() > 1100
    _cmp(oracle, pd, capi.limits(2, 28800, sparse=True), [dense, scans[0]], tag="sparse preset, a few dense rows")
    # support lists of 64 slots: every row overflows into the scan's region, which the preset sizes at 8192 entries
    ctx = capi.Context(p, capi.limits(1, 28800, sparse=True, max_neighbors=64, max_overflow_points=256))
    got = ctx.process_host([scans[0]], roll=0.02, pitch=-0.015)[0]
    ctx.close()
    assert got["flags"] & 0x8 and got["n_keypoints"] == oracle.run(p, scans[0], roll=0.02, pitch=-0.015)["n_keypoints"]


def test_long_support_lists_use_the_workgroup_tiers(fxlib, oracle):
    """Keypoints with > 256 and > 1024 support points (list tier and dense tier)."""
    s = util.vlp16_scan(1000, n_poles=8, x_lo=3.0, x_hi=8.0, y_lo=-4.0, y_hi=4.0)
    p = capi.params("default", descriptor_radius=4.0)
    ora = oracle.run(p, s)
    assert ora["kp_neighbors"].max() > 1100
    _cmp(oracle, p, capi.limits(1, 28800), [s], tag="dense neighbourhoods")


@pytest.mark.parametrize("empty_at", ["first", "middle", "last", "all"])
def test_device_resident_empty_scans_with_null_pointer(fxlib, oracle, empty_at):
    """FX_IN_DEVICE + n_points = 0 + points = NULL is legal (fx_process_batch) and loads nothing."""
    import torch
    p = capi.params("launch")
    full = [util.vlp16_scan(1000 + b) for b in range(2)]
    layout = {"first": [None, 0, 1], "middle": [0, None, 1], "last": [0, 1, None], "all": [None, None, None]}[empty_at]
    dev = [torch.from_numpy(s).cuda() for s in full]
    ctx = capi.Context(p, capi.limits(len(layout), 28800))
    ptrs = [0 if i is None else dev[i].data_ptr() for i in layout]
    cnts = [0 if i is None else 28800 for i in layout]
    descs = ctx.make_descs(ptrs, cnts, 16, 0.02, -0.015)
    for _ in range(2):  # (twice: the second call sees the first one's counters and lists)
        v = ctx.process_raw(descs, len(layout), capi.FX_IN_DEVICE | capi.FX_OUT_HOST | capi.FX_OUT_CLOUDS | capi.FX_OUT_DEBUG)
        got = ctx.unpack(v)
    for b, i in enumerate(layout):
        if i is None:
            assert got[b]["n_keypoints"] == 0 and got[b]["flags"] == 0 and len(got[b]["filtered"]) == 0
        else:
            util.compare_scan(got[b], oracle.run(p, full[i], roll=0.02, pitch=-0.015), tag=f"empty {empty_at}, scan {b}")
    ctx.close()


def test_small_max_points_with_an_empty_last_scan(fxlib, oracle):
    """max_points below one k_prep tile (2048 points) and an empty scan in the last staging slot."""
    s = util.vlp16_scan(1000)[:1500]
    scans = [s, s[:700], np.zeros((0, 4), np.float32)]
    got = _cmp(oracle, capi.params("launch"), capi.limits(3, 1500), scans, 0.02, -0.015, "small max_points")
    assert got[2]["n_keypoints"] == 0


def test_neighbour_a_tenth_of_a_millimetre_from_the_keypoint_counts(fxlib, oracle):
    """3DSC only skips a neighbour whose squared distance is below FLT_MIN (pcl::utils::equal's default
    tolerance is numeric_limits<float>::min(), not epsilon()): a surface point 0.1 mm from a keypoint
    (d2 = 1e-8 < FLT_EPSILON) is binned like any other."""
    p = capi.params("launch")
    s = util.vlp16_scan(1000)
    base = oracle.run(p, s)
    kp = base["keypoints"][:4, :3].astype(np.float64)
    extra = np.zeros((len(kp), 4), np.float32)
    extra[:, :3] = (kp + np.array([1.0e-4, 0.0, 0.0])).astype(np.float32)
    s2 = np.concatenate([s, extra])
    ora = oracle.run(p, s2, want_rotated=True)
    # at least one keypoint now has a neighbour with 0 < d2 < FLT_EPSILON in the oracle's own cloud
    rot = ora["rotated"][:, :3].astype(np.float32)
    near = 0
    for k in ora["keypoints"][:, :3]:
        d = rot - k.astype(np.float32)
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        near += int(((d2 > 0) & (d2 < np.finfo(np.float32).eps)).sum())
    assert near > 0
    ctx = capi.Context(p, capi.limits(1, len(s2)))
    got = ctx.process_host([s2])[0]
    util.compare_scan(got, ora, tag="near-origin neighbour")
    ctx.close()


def test_batches_in_flight_hint_changes_no_result(fxlib, oracle):
    """fx_set_batches_in_flight is a launch-policy hint (the grid-stride kernels take smaller grids when the caller keeps
    several contexts busy): every value gives the same bits, through plain launches and through the graph replay of small batches."""
    p = capi.params("launch")
    scans = [util.vlp16_scan(1000 + b) for b in range(5)] + [np.zeros((0, 4), np.float32)]
    ctx = capi.Context(p, capi.limits(len(scans), 28800))
    ref = None
    for n, graph in ((1, 0), (3, 0), (4, 0), (8, 0), (4, 16), (1, 16)):
        ctx.set_graph_batch(graph)
        ctx.set_batches_in_flight(n)
        got = ctx.process_host(scans, roll=0.02, pitch=-0.015)
        if ref is None:
            ref = got
            for b, s in enumerate(scans):
                util.compare_scan(got[b], oracle.run(p, s, roll=0.02, pitch=-0.015), tag=f"in-flight hint scan {b}")
        for b in range(len(scans)):
            for key in ("flags", "n_keypoints"):
                assert got[b][key] == ref[b][key]
            for key in ("filtered", "candidates", "kpc", "keypoints", "kp_neighbors", "descriptors"):
                util.assert_bit_equal(got[b][key], ref[b][key], f"hint {n} graph {graph} scan {b} {key}")
    ctx.close()
