"""The oracle against the known-answer constants of SURVEY.md Appendix B (the only external pins
that exist: the reference ships no tests or golden vectors, and PCL cannot be built here)."""
import numpy as np


def test_mt19937_stream_and_xaxes(oracle):
    # B-2: boost::mt19937(12345) -> uniform_01 -> float
    u, f = oracle.sc3d_rng(12)
    assert u[:9].tolist() == [3992670690, 3823185381, 1358822685, 561383553, 789925284, 170765737, 878579710,
                              3549516158, 2438360421]
    np.testing.assert_allclose(f[:3], [0.929616094, 0.890154719, 0.316375554], rtol=0, atol=1e-9)
    exp = {0: (0.722270429, 0.691610694), 1: (0.579289973, 0.815121531), 2: (0.240270138, 0.970706105),
           3: (0.666253746, 0.745725155)}
    for k, (ex, ey) in exp.items():
        a, b = np.float32(f[3 * k]), np.float32(f[3 * k + 1])
        n = np.sqrt(np.float32(a * a + np.float32(b * b)))
        assert abs(a / n - ex) < 1e-7 and abs(b / n - ey) < 1e-7


def test_sc3d_tables_R25(oracle):
    radii, theta, phi, lut = oracle.sc3d_tables(2.5)
    exp = [0.25, 0.291478604, 0.339839101, 0.396223307, 0.461962461, 0.53860867, 0.62797159, 0.732161164,
           0.853637278, 0.995267987, 1.16039729, 1.35292387, 1.57739341, 1.83910573, 2.14423966, 2.5]
    np.testing.assert_allclose(radii, exp, rtol=2e-7, atol=0)
    assert radii[15] == np.float32(2.5) and theta[11] == np.float32(180.0) and phi[12] == np.float32(360.0)
    assert abs(theta[1] - 16.363636) < 1e-5 and phi[1] == np.float32(30.0)

    def L(j, k, l=0):
        return lut[l * 165 + k * 15 + j]
    np.testing.assert_allclose([L(0, 0), L(0, 5), L(7, 3), L(14, 5), L(14, 10)],
                               [24.92099, 13.011137, 4.70622826, 1.51698685, 2.90557194], rtol=3e-7)
    assert abs(lut.min() - 1.51699) < 1e-4 and abs(lut.max() - 24.921) < 1e-3
    for l in range(1, 12):  # identical for every azimuth bin
        assert (lut[l * 165:(l + 1) * 165] == lut[:165]).all()


def test_sc3d_tables_R20(oracle):
    radii, _, _, lut = oracle.sc3d_tables(2.0)
    np.testing.assert_allclose([radii[1], radii[14]], [0.233182877, 1.71539176], rtol=3e-7)
    np.testing.assert_allclose([lut[0], lut[5 * 15 + 14]], [31.1512394, 1.89623368], rtol=3e-7)


def test_radius_thresholds(oracle):
    lib = oracle.load()
    # A.4: EuclideanClusterExtraction narrows the tolerance to float, 3DSC passes doubles
    assert lib.fxo_cluster_radius2(0.65) == np.float32(0.422499955)
    assert lib.fxo_cluster_radius2(1.0) == np.float32(1.0)
    assert lib.fxo_cluster_radius2(0.15) == np.float32(0.0225000009)
    assert lib.fxo_cluster_radius2(0.2) == np.float32(0.0400000028)
    assert lib.fxo_radius2(2.5) == np.float32(6.25) and lib.fxo_radius2(0.5) == np.float32(0.25)
    assert lib.fxo_radius2(2.0) == np.float32(4.0) and lib.fxo_radius2(0.4) == np.float32(0.159999996)


def test_sort_tie_order_is_libstdcxx(oracle):
    # B-1: <= 16 clusters: descending, ties keep discovery order; above that they do not
    for n in (8, 16):
        assert oracle.sort_by_size_desc(np.full(n, 3)).tolist() == list(range(n))
    for n in (17, 40, 200):
        p = oracle.sort_by_size_desc(np.full(n, 3)).tolist()
        assert sorted(p) == list(range(n)) and p != list(range(n))
    s = np.array([5, 1, 9, 9, 2, 5], np.uint32)
    assert oracle.sort_by_size_desc(s).tolist() == [2, 3, 0, 5, 4, 1]


def test_elevation_formula_variants(oracle):
    # B-5: the reference's cos/sin formula and atan2(z, hypot) agree after rounding to float
    rng = np.random.default_rng(5)
    lib = oracle.load()
    pts = rng.uniform(-80, 80, (20000, 3)).astype(np.float32)
    a = np.array([lib.fxo_elevation_deg(float(x), float(y), float(z)) for x, y, z in pts], np.float32)
    b = np.degrees(np.arctan2(pts[:, 2].astype(np.float64), np.hypot(pts[:, 0].astype(np.float64),
                                                                     pts[:, 1].astype(np.float64)))).astype(np.float32)
    assert (a == b).all()
    assert lib.fxo_elevation_deg(0.0, 0.0, 0.0) == 0.0
    assert lib.fxo_elevation_deg(0.0, 0.0, 1.0) == 90.0


def test_rotation_identity_and_known(oracle):
    assert oracle.rotation(0.0, 0.0).tolist() == [1, 0, 0, 0, 1, 0, 0, 0, 1]
    R = oracle.rotation(0.02, -0.015).reshape(3, 3).astype(np.float64)
    np.testing.assert_allclose(R @ R.T, np.eye(3), atol=2e-7)
    cy, sy, cx, sx = np.cos(-0.015), np.sin(-0.015), np.cos(0.02), np.sin(0.02)
    exp = np.array([[cy, sy * sx, sy * cx], [0, cx, -sx], [-sy, cy * sx, cy * cx]])  # Ry(pitch) Rx(roll)
    np.testing.assert_allclose(R, exp, atol=2e-7)


def test_secondary_merge_geometry(oracle):
    # B-3: 0.75*0.15/2 per degree => 0.1125 m between adjacent 2-degree rings; xy link distance < 0.0992157
    pz = np.float32(np.float64(np.float32(2.0)) * 0.75 * 0.15 / 2)
    assert abs(float(pz) - 0.1125) < 1e-7
    r2 = oracle.load().fxo_cluster_radius2(0.15)
    assert abs(np.sqrt(r2 - pz * pz) - 0.0992157) < 1e-6
