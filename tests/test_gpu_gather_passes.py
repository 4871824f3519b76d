"""More keypoints in a scan than k_gather's LDS tables hold (FX_GATHER_KCAP = 2048; the limit used to be ~2300 keypoints a scan,
refused at fx_create): the support gather then runs in passes over the keypoints (k_gather_passes), the has-a-neighbour flags of
all passes meet in HBM and the 3DSC x-axis ordinals are dealt over all of them.  A scene of ~3000 small
clusters, each its own keypoint, unflagged and equal to the oracle."""
import numpy as np
import pytest

from feature_extraction_amd import capi
from tests import util
from tests.test_gpu_ring_run_tier import ring_points

pytestmark = pytest.mark.gpu


def many_small_clusters(two_returns_from=None):
    """Clusters of three returns, 0.95 m apart along range shells of rings 6 .. 12 (as far as the filter's z limits let a ring
    reach), every other ring shifted by half a spacing: no two candidates merge (ref: node.cpp:217-229), each is a keypoint."""
    parts = []
    for ring in range(6, 13):
        el = np.radians(-15.0 + 2.0 * ring)
        az, rg = [], []
        for shell in (20.0, 25.0, 30.0, 35.0, 40.0, 45.0):
            z = shell * np.sin(el)
            if not -1.4 < z < 3.9:
                continue
            step = np.degrees(0.95 / shell)
            for c in np.arange(-86.0 + (0.5 * step if ring % 2 else 0.0), 86.0, step):
                if two_returns_from is not None and c >= two_returns_from:  # (the two outer returns only)
                    az += [c, c + 0.2]
                    rg += [shell] * 2
                    continue
                az += list(c + 0.1 * np.arange(3))
                rg += [shell] * 3
        order = np.argsort(np.array(az), kind="stable")
        parts.append(ring_points(ring, np.array(az)[order], np.array(rg)[order]))
    return np.concatenate(parts)


def test_more_keypoints_than_the_gather_tables_hold(fxlib, oracle):
    s = many_small_clusters()
    p = capi.params("launch", number_detection_channels=1, cluster_tolerance=0.65)
    lim = capi.limits(2, 28800, max_ring_points=2400, max_ring_candidates=1024, max_candidates=4096, max_keypoints=4096,
                      max_total_keypoints=8192, max_kpc_points=16384)
    ora = oracle.run(p, s)
    assert ora["n_keypoints"] > 2300, ora["n_keypoints"]  # (beyond the old ceiling, and beyond one pass of 2048)
    ctx = capi.Context(p, lim)
    small = util.vlp16_scan(1000)
    got = ctx.process_host([s, small])
    util.compare_scan(got[0], ora, tag="many keypoints")
    util.compare_scan(got[1], oracle.run(p, small), tag="beside it")
    ctx.close()
    # Keypoints without a neighbour draw no x-axis and shift the ordinals of all later ones — across the
    # passes too: a descriptor radius of 5 cm, and the later clusters reduced to their two outer returns (the centroid of two
    # returns 10 - 16 cm apart has no neighbour; of two 7 - 9 cm apart it has two).
    thin = many_small_clusters(two_returns_from=-40.0)  # (the clusters from azimuth -40 degrees on lose their middle return)
    p2 = capi.params("launch", number_detection_channels=1, cluster_tolerance=0.65, descriptor_radius=0.05)
    o2 = oracle.run(p2, thin)
    lonely = int((o2["kp_neighbors"] == 0).sum())
    assert o2["n_keypoints"] > 2300 and 200 < lonely < o2["n_keypoints"] - 200, (o2["n_keypoints"], lonely)
    ctx = capi.Context(p2, lim)
    util.compare_scan(ctx.process_host([thin])[0], o2, tag="keypoints without neighbours")
    ctx.close()
