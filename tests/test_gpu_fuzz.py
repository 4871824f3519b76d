"""Seeded differential fuzz: random scenes x random node parameters x input perturbations, the C-ABI
against the oracle (integer-exact detector, descriptors within DESC_TOL).  Every case is reproducible
from its seed; the oracle runs the kd-tree search, so a case costs a fraction of a second."""
import numpy as np
import pytest

from feature_extraction_amd import capi
from tests import util

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(seed)
    scene = dict(n_poles=int(rng.integers(0, 160)), pole_radius=float(rng.uniform(0.03, 0.4)),
                 pole_height=float(rng.uniform(0.5, 6.0)), x_lo=float(rng.uniform(-40, 5)), x_hi=float(rng.uniform(10, 70)),
                 y_lo=float(rng.uniform(-40, -5)), y_hi=float(rng.uniform(5, 40)), sensor_height=float(rng.uniform(1.0, 2.5)),
                 wall_radius=float(rng.uniform(40, 120)))
    s = util.vlp16_scan(int(rng.integers(1, 1 << 30)), **scene)
    over = dict(cluster_tolerance=float(rng.uniform(0.2, 1.3)), cluster_min_count=int(rng.integers(1, 6)),
                cluster_max_count=int(rng.integers(20, 1200)), cluster_radius_threshold=float(rng.uniform(0.08, 0.5)),
                number_detection_channels=int(rng.integers(1, 5)), descriptor_radius=float(rng.uniform(0.3, 3.0)),
                x_min=float(rng.uniform(-30, 0)), x_max=float(rng.uniform(30, 100)), y_min=float(rng.uniform(-60, -10)),
                y_max=float(rng.uniform(10, 60)), z_min=float(rng.uniform(-2.5, -0.5)), z_max=float(rng.uniform(1.0, 6.0)),
                cloud_leveling=int(rng.integers(0, 2)))
    p = capi.params("default", **over)
    kind = int(rng.integers(0, 5))
    if kind == 1:    # drop-outs: a tenth of the returns missing (NaN), as a real driver reports them
        s[rng.choice(len(s), len(s) // 10, replace=False), :3] = np.nan
    elif kind == 2:  # range noise of a few centimetres
        r = 1.0 + rng.normal(0.0, 0.004, len(s)).astype(np.float32)
        s[:, :3] *= r[:, None]
    elif kind == 3:  # a ragged, shorter scan (packets lost at the end) with a block of duplicates
        s = s[: int(len(s) * rng.uniform(0.3, 0.95))].copy()
        s[100:160] = s[40:100]
    elif kind == 4:  # azimuth blocks out of order (driver packets reordered)
        blocks = np.array_split(np.arange(len(s)), 40)
        s = s[np.concatenate([blocks[i] for i in rng.permutation(len(blocks))])]
    roll, pitch = float(rng.uniform(-0.08, 0.08)), float(rng.uniform(-0.08, 0.08))
    return np.ascontiguousarray(s), p, roll, pitch, dict(scene=scene, over=over, kind=kind)


@pytest.mark.parametrize("path", ["front", "front-fused", "front-split", "front-split-lean", "separate", "separate-large-merge", "separate-large-merge-sliced", "front-redo", "front-tail"])
@pytest.mark.parametrize("block", range(8))
def test_random_scenes_parameters_and_perturbations(fx_hooks, oracle, block, path):
    """Every case through the front path a scan per call takes by default (the sliced streaming pass and ring split + k_front_cd:
    batches of up to eight scans), through the fused front kernel (k_front: what larger batches take), through its two-launch forms (k_front_ab + k_front_cd / k_front_cdl:
    measured in round 6, not the default), through the separate kernels, through those with the LDS merge
    tier's capacity lowered (scans with more than 16 candidates take the large merge tier: as one launch, and as its three
    launches with five workgroups a scan in the pair loop), and through the two kernels behind
    k_front (k_front_redo: the general bodies in k_front's shape; k_slow: the same on scratch in HBM)."""
    if path in ("front-redo", "front-tail", "separate-large-merge-sliced", "front-split") and block >= 4:
        pytest.skip("half of the blocks are enough for the rarely used kernels")
    fx_hooks(**{"front": {}, "front-fused": dict(FX_FRONT_STREAM=0), "front-split": dict(FX_FRONT_SPLIT=1), "front-split-lean": dict(FX_FRONT_SPLIT=2), "separate": dict(FX_FRONT=0), "separate-large-merge": dict(FX_FRONT=0, FX_MERGE_BIG_CAP=16, FX_MERGE_SLICES=1),
                "separate-large-merge-sliced": dict(FX_FRONT=0, FX_MERGE_BIG_CAP=16, FX_MERGE_SLICES=5), "front-redo": dict(FX_FRONT_FORCE=1), "front-tail": dict(FX_FRONT_FORCE=2)}[path])
    total_k = 0
    for seed in range(block * 10, block * 10 + 10):
        s, p, roll, pitch, what = _case(seed)
        ctx = capi.Context(p, capi.limits(1, 28800, max_candidates=3500, max_kpc_points=57600, max_keypoints=1024, max_total_keypoints=1024,
                                          max_ring_candidates=512))
        got = ctx.process_host([s], roll=roll, pitch=pitch)[0]
        ora = oracle.run(p, s, roll=roll, pitch=pitch)
        st = util.compare_scan(got, ora, tag=f"fuzz seed {seed} {what}")
        total_k += st["K"]
        ctx.close()
    assert total_k > 0, "a whole block of cases without a single keypoint tests nothing"


def test_unstructured_clouds_are_flagged_or_exact_never_fatal(fxlib, oracle):
    """Points with arbitrary elevations and no scan order (nothing like a spinning LiDAR): rings overflow
    their capacity, candidates overflow theirs — the call must come back with flags, and whatever is not
    flagged must still match the oracle."""
    rng = np.random.default_rng(5)
    for n, B in ((28800, 4), (6000, 16), (20000, 8)):
        scans = []
        for _ in range(B):
            pts = np.zeros((n, 4), np.float32)
            pts[:, 0] = rng.uniform(0, 100, n)
            pts[:, 1] = rng.uniform(-50, 50, n)
            pts[:, 2] = rng.uniform(-1.5, 4, n)
            scans.append(pts)
        for preset in ("launch", "default"):
            p = capi.params(preset)
            ctx = capi.Context(p, capi.limits(B, 28800))
            got = ctx.process_host(scans)
            for b in range(2):
                if got[b]["flags"] == 0:
                    util.compare_scan(got[b], oracle.run(p, scans[b]), tag=f"unstructured {n} {preset} {b}")
            ctx.close()
