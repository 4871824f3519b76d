"""Shared helpers of the parity tests: scene builders and the product-vs-oracle comparison."""
import numpy as np

from feature_extraction_amd import capi

DESC_TOL = 1e-5  # north_star: descriptor values within 1e-5 (absolute, on values of O(1..100))


def vlp16_scan(seed, **over):
    return capi.synth_scan(capi.synth_cfg(seed, **over))


def golden_case(g, name):
    """(params, limits, scan, roll, pitch) of a tests/golden fixture.  The VLP-16 fixtures store their input; the
    64- and 128-ring ones store the generator's configuration ("spec") and a checksum of the scan it makes."""
    import json
    if "spec" in g.files:
        c = json.loads(bytes(g["spec"]).decode())
        pts = capi.synth_scan(capi.synth_cfg(c["seed"], **c["synth"]))
        assert int(pts.view(np.uint32).astype(np.uint64).sum()) == int(g["points_checksum"][0]), "the generator no longer makes the fixture's scan"
        return capi.params(c["preset"], **c["params"]), capi.limits(1, len(pts), **c["limits"]), pts, c["roll"], c["pitch"]
    preset = "default" if "default" in name else "launch"
    _, roll, pitch = g["meta"]
    pts = np.concatenate([g["points_xyz"], np.zeros((len(g["points_xyz"]), 1), np.float32)], axis=1)
    return capi.params(preset), capi.limits(2, 28800), pts, float(roll), float(pitch)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    if a.dtype.kind == "f":
        bad = bits(a) != bits(b)
        # +0 / -0 are the same value for every consumer; NaN payloads must still agree in NaN-ness
        bad &= ~((a == 0) & (b == 0))
    else:
        bad = a != b
    if bad.any():
        idx = np.argwhere(bad)[:5]
        raise AssertionError(f"{what}: {int(bad.sum())} of {bad.size} differ, first at {idx.tolist()}: "
                             f"{a[tuple(idx[0])]!r} vs {b[tuple(idx[0])]!r}")


def compare_scan(got, ora, estimate_descriptors=True, tag=""):
    """got: one dict from Context.process_host(debug=True); ora: oracle_py.run() of the same scan.
    Integer-exact on membership / order, bit-exact on every float the detector emits,
    descriptors within DESC_TOL.  Returns stats for reporting."""
    assert got["flags"] == 0, f"{tag} flags {got['flags']:#x}"
    assert_bit_equal(got["filtered"], ora["filtered"], f"{tag} filtered cloud")
    assert_bit_equal(got["candidates"], ora["candidates"], f"{tag} keypoints_full")
    assert_bit_equal(got["cand_size"], ora["cand_size"], f"{tag} per-ring cluster sizes")
    assert_bit_equal(got["kpc"], ora["kpc"], f"{tag} keypoint_cloud (cluster membership)")
    assert_bit_equal(got["kpc_cand"], ora["kpc_cand"], f"{tag} keypoint_cloud -> candidate")
    assert_bit_equal(got["cand_keypoint"], ora["cand_keypoint"], f"{tag} candidate -> keypoint (merge membership)")
    assert got["n_keypoints"] == ora["n_keypoints"], f"{tag} K {got['n_keypoints']} vs {ora['n_keypoints']}"
    assert_bit_equal(got["keypoints"], ora["keypoints"], f"{tag} keypoints")
    assert_bit_equal(got["kp_size"], ora["kp_size"], f"{tag} keypoint sizes")
    stats = {"K": int(got["n_keypoints"]), "max_abs": 0.0, "n_inexact": 0, "n_values": 0}
    if estimate_descriptors and got["n_keypoints"]:
        assert_bit_equal(got["kp_neighbors"], ora["kp_neighbors"], f"{tag} 3DSC neighbour counts")
        g, o = got["descriptors"], ora["descriptors"]
        assert g.shape == o.shape
        nan_g, nan_o = np.isnan(g), np.isnan(o)
        assert (nan_g == nan_o).all(), f"{tag} NaN pattern of descriptors differs"
        diff = np.abs(np.where(nan_g, 0, g) - np.where(nan_o, 0, o))
        stats["max_abs"] = float(diff.max()) if diff.size else 0.0
        stats["n_inexact"] = int((diff != 0).sum())
        stats["n_values"] = int(diff.size)
        assert stats["max_abs"] <= DESC_TOL, f"{tag} descriptor max |diff| {stats['max_abs']} > {DESC_TOL}"
        assert (g[:, 1980:] == 0).all(), f"{tag} rf must be zero"
    return stats
