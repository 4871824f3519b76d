"""Host-side pieces of the product library (no GPU needed): parameter presets, the rotation
matrix, 3DSC tables and x-axes, the synthetic generator, sharding and record packing."""
import ctypes as C

import numpy as np

from feature_extraction_amd import capi, sharding


def test_default_preset_is_the_node_constructor(fxlib):
    p = capi.params("default")  # ref: src/feature_extraction_node.cpp:9-34
    assert (p.cloud_leveling, p.x_min, p.x_max, p.y_min, p.y_max, p.z_min, p.z_max) == (1, 0.0, 75.0, -30.0, 30.0, -1.5, 5.0)
    assert (p.cluster_tolerance, p.cluster_min_count, p.cluster_max_count) == (0.65, 5, 50)
    assert (p.cluster_radius_threshold, p.number_detection_channels) == (0.15, 1)
    assert (p.estimate_descriptors, p.descriptor_radius) == (1, 2.5)
    assert (p.n_rings, p.el0_deg, p.el_step_deg, p.secondary_max) == (16, -15.0, 2.0, 16)  # ref: :195, :200, :227
    for i in range(16):  # ref: node.cpp:200  (i-7)*2-1
        assert p.el0_deg + i * p.el_step_deg == (i - 7) * 2 - 1


def test_launch_preset_is_the_launch_file(fxlib):
    p = capi.params("launch")  # ref: launch/keypoint_playback.launch:17-33
    assert (p.cluster_tolerance, p.cluster_min_count, p.cluster_max_count) == (1.0, 1, 1000)
    assert (p.cluster_radius_threshold, p.number_detection_channels) == (0.2, 2)
    assert (p.x_min, p.x_max, p.y_min, p.y_max, p.z_min, p.z_max) == (0.0, 100.0, -50.0, 50.0, -1.5, 4.0)
    assert p.descriptor_radius == 2.5 and p.estimate_descriptors == 1


def test_rotation_matches_oracle(fxlib, oracle):
    rng = np.random.default_rng(3)
    for roll, pitch in [(0, 0), (0.02, -0.015), (np.pi, 0.3), (-3.1, 0.01)] + [tuple(rng.uniform(-3.2, 3.2, 2)) for _ in range(200)]:
        R = np.zeros(9, np.float32)
        fxlib.fx_rotation_from_roll_pitch(float(roll), float(pitch), R.ctypes.data_as(capi._F32P))
        assert R.view(np.uint32).tolist() == oracle.rotation(float(roll), float(pitch)).view(np.uint32).tolist()


def test_tables_and_xaxes_match_oracle(fxlib, oracle):
    for Rd in (2.5, 2.0, 1.0, 7.3):
        mine = [np.zeros(k, np.float32) for k in (16, 12, 13, 1980)]
        fxlib.fx_sc3d_tables(Rd, *[m.ctypes.data_as(capi._F32P) for m in mine])
        for a, b in zip(mine, oracle.sc3d_tables(Rd)):
            assert a.view(np.uint32).tolist() == b.view(np.uint32).tolist()
    _, f = oracle.sc3d_rng(3 * 300)
    for k in list(range(40)) + [255, 299]:
        xy = np.zeros(2, np.float32)
        fxlib.fx_sc3d_xaxis(k, xy.ctypes.data_as(capi._F32P))
        a, b = np.float32(f[3 * k]), np.float32(f[3 * k + 1])
        n = np.sqrt(np.float32(a * a + np.float32(b * b + np.float32(0.0))))
        assert xy[0] == np.float32(a / n) and xy[1] == np.float32(b / n)


def test_synthetic_generator_shape(fxlib):
    cfg = capi.synth_cfg(1000)
    pts = capi.synth_scan(cfg)
    assert pts.shape == (28800, 4) and np.isfinite(pts).all() and (pts[:, 3] == 0).all()
    assert np.array_equal(pts, capi.synth_scan(capi.synth_cfg(1000)))          # reproducible
    assert not np.array_equal(pts, capi.synth_scan(capi.synth_cfg(1001)))      # seeded
    # firing order: azimuth-major, ring-minor; every ray returns
    el = np.degrees(np.arctan2(pts[:, 2], np.hypot(pts[:, 0], pts[:, 1]))).reshape(1800, 16)
    np.testing.assert_allclose(el, np.tile(-15 + 2 * np.arange(16), (1800, 1)), atol=1e-3)
    rng_xy = np.hypot(pts[:, 0], pts[:, 1])
    assert rng_xy.max() <= 90.0 + 1e-3
    assert ((pts[:, 2] > -1.8 - 1e-4)).all()
    # poles exist: some returns closer than ground/wall would be
    assert (rng_xy.reshape(1800, 16)[:, 8:] < 80).sum() > 50


def test_limits_defaults(fxlib):
    l = capi.limits(1024, 28800)
    assert (l.max_batch, l.max_points, l.max_ring_points, l.max_ring_candidates, l.max_candidates, l.max_keypoints,
            l.max_neighbors, l.max_total_keypoints, l.max_kpc_points, l.max_dense_points) == (
                1024, 28800, 2048, 256, 2048, 256, 1024, 65536, 4096, 1024 * 28800)
    assert capi.limits(64, 128 * 2048).max_neighbors == 4096  # dense many-ring scans: longer support lists
    assert capi.limits(1, 28800).max_dense_points == 32 * 28800  # a one-scan context still holds a scan of overlapping support sets
    assert (l.max_overflow_points, capi.limits(1, 28800).max_overflow_points, capi.limits(8, 1000).max_overflow_points) == (28800, 32 * 28800, 4000)
    # the sparse preset (VLP-16-class workloads): only the dense tier's pools and the overflow regions shrink
    s = capi.limits(1024, 28800, sparse=True)
    assert (s.max_dense_points, s.max_overflow_points) == (262144, 8192)
    assert all(getattr(s, f) == getattr(l, f) for f, _ in capi.FxLimits._fields_ if f not in ("max_dense_points", "max_overflow_points"))
    assert capi.limits(4, 131072, sparse=True).max_dense_points == 4 * 131072 and capi.limits(1, 1000, sparse=True).max_overflow_points == 1000


def test_shard_plan_is_a_partition():
    for total in (0, 1, 7, 1024, 8191):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(e - s for s, e in spans) - min(e - s for s, e in spans) <= 1
            for scan in range(0, total, max(1, total // 50)):
                r = sharding.owner_of(scan, total, world)
                assert spans[r][0] <= scan < spans[r][1]


def test_record_roundtrip():
    rng = np.random.default_rng(0)
    kps = [rng.normal(size=(k, 4)).astype(np.float32) for k in (0, 1, 5, 127, 130)]
    flags = [0, 0, 2, 0, 0]
    rec = sharding.pack_records(kps, flags)
    assert rec.shape == (5, 128, 4) and rec.nbytes == 5 * 2048
    got = sharding.unpack_records(rec)
    for (k, f, a), kp, fl in zip(got, kps, flags):
        assert k == min(len(kp), 127) and np.array_equal(a, kp[:k])
        assert f == (fl | (4 if len(kp) > 127 else 0))
