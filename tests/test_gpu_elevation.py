"""k_prep's table-driven elevation angle against the library path and the oracle (ref: node.cpp:147-156).

The fast path may only ever decline a point (fast_ok = 0: the point takes the exact path); a value it vouches for must be
the float the reference computes."""
import ctypes as C

import numpy as np
import pytest

from feature_extraction_amd import capi

pytestmark = pytest.mark.gpu


def _device(xyz):
    lib = capi.load_test()  # (the fx_test_* entry points exist only in the test build)
    xyz = np.ascontiguousarray(xyz, np.float32)
    n = len(xyz)
    fast, exact, ok = np.empty(n, np.float32), np.empty(n, np.float32), np.empty(n, np.uint8)
    capi.check(lib.fx_test_elevation_device(0, xyz.ctypes.data, n, fast.ctypes.data, ok.ctypes.data, exact.ctypes.data))
    return fast, ok.astype(bool), exact


def _lidar_points(rng, n):
    rho = rng.uniform(0.3, 120.0, n)
    az = rng.uniform(-np.pi, np.pi, n)
    el = np.deg2rad(rng.uniform(-32.0, 32.0, n))
    return np.stack([rho * np.cos(el) * np.cos(az), rho * np.cos(el) * np.sin(az), rho * np.sin(el)], 1).astype(np.float32)


def test_fast_path_agrees_with_the_exact_path_on_ten_million_points():
    rng = np.random.default_rng(7)
    pts = _lidar_points(rng, 10_000_000)
    fast, ok, exact = _device(pts)
    assert ok.mean() > 0.999, f"the fast path declines {1 - ok.mean():.2%} of ordinary points"
    assert 100 < (~ok).sum() < 3000  # (2^-14 of them lie next to a rounding midpoint: about 600)
    bad = ok & (fast.view(np.uint32) != exact.view(np.uint32))
    assert not bad.any(), f"{int(bad.sum())} vouched-for values differ, first {pts[bad][0]!r}: {fast[bad][0]!r} vs {exact[bad][0]!r}"


def test_fast_path_against_the_oracle():
    from oracle import oracle_py
    olib = oracle_py.load()
    rng = np.random.default_rng(11)
    pts = _lidar_points(rng, 200_000)
    fast, ok, exact = _device(pts)
    ref = np.array([olib.fxo_elevation_deg(float(x), float(y), float(z)) for x, y, z in pts], np.float32)
    assert (exact.view(np.uint32) == ref.view(np.uint32)).all()
    assert (fast.view(np.uint32)[ok] == ref.view(np.uint32)[ok]).all()


def test_points_the_expansion_does_not_cover_are_declined():
    cases = np.array([
        [0, 0, 0], [0, 0, 1], [0, 0, -1],            # |xy| = 0
        [1, 0, 2], [1, 1, -5], [0.1, 0, 0.1001],     # steeper than 45 degrees
        [1e-20, 0, 1e-21], [1e20, 1e20, 1e19],       # |xy|^2 outside the fp32 seed's range
        [1, 0, 1e-42], [3, 4, 1e-38],                # results below the normal floats
        [np.nan, 0, 1], [1, np.nan, 0], [1, 0, np.nan], [np.inf, 0, 1], [1, 0, np.inf],
    ], np.float32)
    fast, ok, exact = _device(cases)
    assert not ok.any(), f"vouched for {cases[ok]!r}"
    # covered corner: z = +-0 gives +-0, t exactly 1 is inside the table
    fast, ok, exact = _device(np.array([[1, 2, 0.0], [1, 2, -0.0], [3, 4, 5], [3, 4, -5]], np.float32))
    assert ok[:2].all() and (fast[:2] == 0).all() and (exact[:2] == 0).all()
    assert (fast[2:].view(np.uint32)[ok[2:]] == exact[2:].view(np.uint32)[ok[2:]]).all()


def test_every_float_z_of_a_stretch_at_fixed_xy():
    # consecutive floats z at fixed |xy| = 1: neighbouring elevations, many of them one or two float values apart
    z = np.arange(np.float32(0.2679).view(np.uint32), np.float32(0.2679).view(np.uint32) + 2_000_000, dtype=np.uint32).view(np.float32)
    pts = np.stack([np.full_like(z, 0.6), np.full_like(z, 0.8), z], 1)
    fast, ok, exact = _device(pts)
    assert (fast.view(np.uint32)[ok] == exact.view(np.uint32)[ok]).all()
    assert (~ok).mean() < 0.01
