"""GPU parity: the HIP path through the C-ABI vs the CPU oracle on identical inputs.
Integer-exact membership / order / counts, bit-exact detector floats, descriptors within 1e-5."""
import glob
import os

import numpy as np
import pytest

from feature_extraction_amd import capi
from tests import util

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def _run_and_compare(oracle, p, lim, scans, roll, pitch, tag, descriptors=True):
    ctx = capi.Context(p, lim)
    got = ctx.process_host(scans, roll=roll, pitch=pitch)
    tot = {"K": 0, "max_abs": 0.0, "n_inexact": 0}
    for b, s in enumerate(scans):
        ro = roll[b] if np.ndim(roll) else roll
        pi = pitch[b] if np.ndim(pitch) else pitch
        ora = oracle.run(p, s, roll=ro, pitch=pi)
        st = util.compare_scan(got[b], ora, estimate_descriptors=descriptors, tag=f"{tag} scan {b}")
        tot["K"] += st["K"]
        tot["max_abs"] = max(tot["max_abs"], st["max_abs"])
        tot["n_inexact"] += st["n_inexact"]
    ctx.close()
    print(f"\n[{tag}] scans {len(scans)} K {tot['K']} descriptor max|diff| {tot['max_abs']:.3g} inexact {tot['n_inexact']}")
    return tot


@pytest.mark.parametrize("preset", ["default", "launch"])
@pytest.mark.parametrize("leveled", [False, True])
def test_vlp16_batch_matches_oracle(fxlib, oracle, preset, leveled):
    B = 6
    scans = [util.vlp16_scan(1000 + b) for b in range(B)]
    roll, pitch = (0.02, -0.015) if leveled else (0.0, 0.0)
    tot = _run_and_compare(oracle, capi.params(preset), capi.limits(B, 28800), scans, roll, pitch, f"{preset} leveled={leveled}")
    assert tot["K"] > 0


def test_per_scan_roll_pitch_and_more_poles(fxlib, oracle):
    B = 5
    rng = np.random.default_rng(2)
    scans = [util.vlp16_scan(2000 + b, n_poles=256) for b in range(B)]
    roll = rng.uniform(-0.05, 0.05, B)
    pitch = rng.uniform(-0.05, 0.05, B)
    tot = _run_and_compare(oracle, capi.params("launch"), capi.limits(B, 28800, max_total_keypoints=B * 256), scans, roll, pitch,
                           "256 poles, per-scan attitude")
    assert tot["K"] > 300  # more than 16 keypoints per scan: the unstable tie order of std::sort is in play


def test_inverted_mount_roll_near_pi(fxlib, oracle):
    # ref: node.cpp:65 roll = imu_roll - pi: a large roll is the normal operating point of the node
    scans = [util.vlp16_scan(77)]
    scans[0][:, 1] *= -1
    scans[0][:, 2] *= -1  # sensor upside down
    _run_and_compare(oracle, capi.params("launch"), capi.limits(1, 28800), scans, np.pi - 0.01, 0.005, "inverted mount")


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_golden_fixtures(fxlib, path):
    """Committed fixtures (inputs + oracle outputs): the comparison needs neither the oracle library
    nor /root/reference on the GPU box."""
    g = np.load(path)
    p, lim, pts, roll, pitch = util.golden_case(g, os.path.basename(path))
    ctx = capi.Context(p, lim)
    got = ctx.process_host([pts], roll=float(roll), pitch=float(pitch))[0]
    ora = {k: g[k] for k in g.files}
    ora["n_keypoints"] = len(g["keypoints"])
    st = util.compare_scan(got, ora, tag=os.path.basename(path))
    assert st["K"] == len(g["keypoints"])
    ctx.close()


def test_descriptors_disabled(fxlib, oracle):
    p = capi.params("launch", estimate_descriptors=0)
    scans = [util.vlp16_scan(5)]
    _run_and_compare(oracle, p, capi.limits(1, 28800), scans, 0.0, 0.0, "no descriptors", descriptors=False)


def test_other_radii_and_tolerances(fxlib, oracle):
    scans = [util.vlp16_scan(300 + b) for b in range(3)]
    for over in (dict(descriptor_radius=2.0), dict(descriptor_radius=1.0, cluster_tolerance=0.4),
                 dict(cluster_min_count=3, cluster_max_count=20, cluster_radius_threshold=0.3, number_detection_channels=3),
                 dict(x_min=-75.0, y_min=-75.0, y_max=75.0, z_min=-1.0)):
        _run_and_compare(oracle, capi.params("default", **over), capi.limits(3, 28800), scans, 0.02, -0.015, str(over))


def test_keypoints_without_neighbours_shift_the_rng_stream(fxlib, oracle):
    """A keypoint with no neighbour inside R gets a NaN descriptor and draws no x-axis, so every later
    keypoint's RNG ordinal moves: k_gather knows which keypoints have a neighbour before any descriptor
    is computed and hands every row its x-axis (k_rng_ord when several workgroups of it share a scan)."""
    scans = [util.vlp16_scan(1000 + b) for b in range(3)]
    ora_small = None
    for radius in (0.1, 0.04):
        p = capi.params("launch", descriptor_radius=radius)
        ora = ora_small = [oracle.run(p, s, roll=0.02, pitch=-0.015) for s in scans]
        assert any((o["kp_neighbors"] == 0).any() and (o["kp_neighbors"] > 0).any() for o in ora)
        ctx = capi.Context(p, capi.limits(3, 28800))
        got = ctx.process_host(scans, roll=0.02, pitch=-0.015)
        for b in range(3):
            util.compare_scan(got[b], ora[b], tag=f"rng shift R={radius} scan {b}")
        ctx.close()
    # a batch large enough for k_gather to run one workgroup per scan, which then settles the ordinals itself
    p = capi.params("launch", descriptor_radius=0.04)
    ctx = capi.Context(p, capi.limits(128, 28800))
    got = ctx.process_host([scans[b % 3] for b in range(128)], roll=0.02, pitch=-0.015)
    for b in (0, 1, 2, 64, 127):
        util.compare_scan(got[b], ora_small[b % 3], tag=f"rng shift, one workgroup per scan, scan {b}")
    ctx.close()
