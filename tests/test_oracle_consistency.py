"""The oracle against itself and against independent statements: brute-force vs kd-tree search,
scipy connected components for cluster membership, golden fixtures, numpy re-statements of the
cheap stages, and the libm-float diagnostic mode."""
import glob
import os

import numpy as np
import pytest
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components
from scipy.spatial import cKDTree

from feature_extraction_amd import capi
from tests import util

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
KEYS = ("filtered", "candidates", "cand_size", "cand_keypoint", "kpc", "kpc_cand", "keypoints", "kp_size",
        "kp_neighbors", "descriptors")


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_golden_fixtures_reproduce(oracle, path):
    g = np.load(path)
    p, _lim, pts, roll, pitch = util.golden_case(g, os.path.basename(path))
    r = oracle.run(p, pts, roll=roll, pitch=pitch, search=oracle.SEARCH_KDTREE)
    for k in KEYS:
        util.assert_bit_equal(r[k], g[k], f"{os.path.basename(path)}:{k}")
    if "points_xyz" in g.files:  # the fixture's input is what the product's generator makes for that seed
        assert np.array_equal(util.vlp16_scan(int(g["meta"][0]))[:, :3], g["points_xyz"])


@pytest.mark.parametrize("preset", ["default", "launch"])
def test_kdtree_equals_brute_force(oracle, preset):
    p = capi.params(preset)
    for seed in (7, 8):
        pts = util.vlp16_scan(seed, n_az=600)
        a = oracle.run(p, pts, roll=0.01, pitch=0.02, search=oracle.SEARCH_BRUTE, want_labels=True)
        b = oracle.run(p, pts, roll=0.01, pitch=0.02, search=oracle.SEARCH_KDTREE, want_labels=True)
        for k in KEYS + ("ring_labels",):
            util.assert_bit_equal(a[k], b[k], f"{preset} seed {seed} {k}")


def test_filter_and_elevation_match_numpy(oracle):
    p = capi.params("default")
    pts = util.vlp16_scan(21)
    pts[5] = [np.nan, 1, 1, 0]
    pts[9] = [1, np.inf, 1, 0]
    roll, pitch = 0.03, -0.02
    r = oracle.run(p, pts, roll=roll, pitch=pitch, want_rotated=True)
    R = oracle.rotation(roll, pitch).reshape(3, 3)
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    with np.errstate(invalid="ignore"):
        rot = np.stack([((R[i, 0] * x + R[i, 1] * y) + R[i, 2] * z) + np.float32(0) for i in range(3)], axis=1)
        el = np.degrees(np.arctan2(z.astype(np.float64), np.hypot(x.astype(np.float64), y.astype(np.float64)))).astype(np.float32)
    fin = np.isfinite(rot).all(axis=1)
    util.assert_bit_equal(r["rotated"][fin, :3], rot[fin], "rotated cloud")
    keep = fin & (rot[:, 2] >= np.float32(-1.5)) & (rot[:, 2] <= np.float32(5.0)) & (rot[:, 1] >= -30) & \
        (rot[:, 1] <= 30) & (rot[:, 0] >= 0) & (rot[:, 0] <= 75)
    util.assert_bit_equal(r["filtered"][:, :3], rot[keep], "filtered xyz (stable order)")
    util.assert_bit_equal(r["filtered"][:, 3], el[keep], "elevation stored in intensity")


@pytest.mark.parametrize("preset", ["default", "launch"])
def test_ring_clusters_match_scipy_components(oracle, preset):
    """Independent cross-check of cluster MEMBERSHIP: radius graph from cKDTree.query_pairs +
    scipy connected components, per ring."""
    p = capi.params(preset)
    tol = float(np.float32(p.cluster_tolerance))
    checked = 0
    for seed in (31, 32, 33):
        pts = util.vlp16_scan(seed)
        r = oracle.run(p, pts, roll=0.02, pitch=-0.015, want_labels=True)
        f = r["filtered"].astype(np.float64)
        for ring in range(p.n_rings):
            lab = r["ring_labels"][ring]
            idx = np.where(lab >= 0)[0]
            c = (ring - 7) * 2 - 1
            assert np.array_equal(idx, np.where((r["filtered"][:, 3] >= c - 1) & (r["filtered"][:, 3] <= c + 1))[0])
            if len(idx) < 2:
                continue
            P = f[idx, :3]
            tree = cKDTree(P)
            pairs = tree.query_pairs(tol, output_type="ndarray")
            # skip the rare ring with a pair sitting on the threshold (fp32 d2 < r2 vs fp64 d <= r)
            d = np.linalg.norm(P[pairs[:, 0]] - P[pairs[:, 1]], axis=1) if len(pairs) else np.zeros(0)
            near = tree.query_pairs(tol * (1 + 1e-5), output_type="ndarray")
            if len(near) != len(pairs) or (len(d) and d.max() > tol * (1 - 1e-5)):
                continue
            g = coo_matrix((np.ones(len(pairs)), (pairs[:, 0], pairs[:, 1])), shape=(len(idx), len(idx)))
            ncomp, comp = connected_components(g, directed=False)
            # oracle label = smallest filtered index of the component
            first = np.full(ncomp, np.iinfo(np.int64).max)
            np.minimum.at(first, comp, idx)
            assert np.array_equal(lab[idx], first[comp]), (seed, ring)
            checked += 1
    assert checked >= 30


@pytest.mark.parametrize("name", ["hdl64_64x2048_launch_seed10", "dense_128x2048_R2m_launch_seed10"])
def test_ring_clusters_of_the_many_ring_fixtures_match_scipy_components(oracle, name):
    """The same cross-check on the 64- and 128-ring fixtures (until round 5 VLP-16 scenes only): dense rings of
    up to a thousand points, window edges at non-integer elevations (centre -+ step / 2 narrowed to float as PassThrough does,
    ref: node.cpp:200-201)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    p, _lim, pts, roll, pitch = util.golden_case(g, name)
    tol = float(np.float32(p.cluster_tolerance))
    r = oracle.run(p, pts, roll=roll, pitch=pitch, want_labels=True)
    f = r["filtered"].astype(np.float64)
    el = r["filtered"][:, 3]
    checked = members = 0
    for ring in range(p.n_rings):
        lab = r["ring_labels"][ring]
        idx = np.where(lab >= 0)[0]
        centre = p.el0_deg + ring * p.el_step_deg
        lo, hi = np.float32(centre - p.el_step_deg / 2.0), np.float32(centre + p.el_step_deg / 2.0)
        assert np.array_equal(idx, np.where((el >= lo) & (el <= hi))[0]), ring
        if len(idx) < 2:
            continue
        P = f[idx, :3]
        tree = cKDTree(P)
        pairs = tree.query_pairs(tol, output_type="ndarray")
        d = np.linalg.norm(P[pairs[:, 0]] - P[pairs[:, 1]], axis=1) if len(pairs) else np.zeros(0)
        near = tree.query_pairs(tol * (1 + 1e-5), output_type="ndarray")
        if len(near) != len(pairs) or (len(d) and d.max() > tol * (1 - 1e-5)):
            continue  # (a pair sitting on the threshold: fp32 d2 < r2 against fp64 d <= r)
        gr = coo_matrix((np.ones(len(pairs)), (pairs[:, 0], pairs[:, 1])), shape=(len(idx), len(idx)))
        ncomp, comp = connected_components(gr, directed=False)
        first = np.full(ncomp, np.iinfo(np.int64).max)
        np.minimum.at(first, comp, idx)
        assert np.array_equal(lab[idx], first[comp]), ring
        checked += 1
        members += len(idx)
    assert checked >= p.n_rings // 3 and members > 5000, (checked, members)


def test_keypoints_are_centroids_of_their_members(oracle):
    p = capi.params("launch")
    pts = util.vlp16_scan(41)
    r = oracle.run(p, pts, roll=0.0, pitch=0.0)
    assert r["n_keypoints"] > 10
    for k in range(r["n_keypoints"]):
        mem = np.where(r["cand_keypoint"] == k)[0]
        assert len(mem) == r["kp_size"][k] and 2 <= len(mem) <= 16
        c = r["candidates"][mem, :3].astype(np.float64).sum(axis=0) / len(mem)
        util.assert_bit_equal(r["keypoints"][k, :3], c.astype(np.float32), f"keypoint {k}")
        assert r["keypoints"][k, 3] == r["candidates"][mem.min(), 3]
    # per-ring candidates are centroids of their keypoint_cloud members
    for c in range(len(r["candidates"])):
        mem = r["kpc"][r["kpc_cand"] == c]
        assert len(mem) == r["cand_size"][c]
        cen = mem[:, :3].astype(np.float64).sum(axis=0) / len(mem)
        util.assert_bit_equal(r["candidates"][c, :3], cen.astype(np.float32), f"candidate {c}")
    sizes = r["kp_size"]
    assert (np.diff(sizes.astype(int)) <= 0).all()  # PCL returns clusters largest first


def test_descriptor_mass_and_rf(oracle):
    p = capi.params("default")
    pts = util.vlp16_scan(1000)
    r = oracle.run(p, pts, roll=0.02, pitch=-0.015)
    d = r["descriptors"]
    assert d.shape == (r["n_keypoints"], 1989) and (d[:, 1980:] == 0).all() and (d[:, :1980] >= 0).all()
    assert ((d[:, :1980] > 0).sum(axis=1) <= r["kp_neighbors"]).all()


def test_libm_float_trig_mode_is_close(oracle):
    """Diagnostic: the literal atan2f/acosf of this glibc vs the fp64-rounded policy (A.8-14)."""
    p = capi.params("launch")
    pts = util.vlp16_scan(1000)
    a = oracle.run(p, pts, roll=0.02, pitch=-0.015, trig=oracle.TRIG_F64_ROUNDED)["descriptors"]
    b = oracle.run(p, pts, roll=0.02, pitch=-0.015, trig=oracle.TRIG_LIBM_F32)["descriptors"]
    assert a.shape == b.shape
    moved = int((a != b).sum())
    assert moved <= 0.001 * a.size, moved  # at most a few bin-edge flips in ~100k bins
