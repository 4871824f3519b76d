"""BASELINE.json configs 3 and 5 (more rings, denser scans) at parity-test size, and the
batch-level outputs: feature records, keypoint records, timings."""
import ctypes as C

import numpy as np
import pytest

from feature_extraction_amd import capi, sharding
from tests import util

pytestmark = pytest.mark.gpu


def _hdl64(seed):
    return capi.synth_scan(capi.synth_cfg(seed, n_rings=64, n_az=2048, el0_deg=-24.8, el_step_deg=26.8 / 63, n_poles=256))


def test_hdl64_style_scans(fxlib, oracle):
    """64 x 2048 (BASELINE config 3).  n_rings / secondary_max are build extensions: the reference
    hard-codes 16 (ref: node.cpp:195, 200, 227)."""
    B = 2
    scans = [_hdl64(10 + b) for b in range(B)]
    for preset in ("default", "launch"):
        p = capi.params(preset, n_rings=64, el0_deg=-24.8, el_step_deg=26.8 / 63, secondary_max=64)
        ctx = capi.Context(p, capi.limits(B, 64 * 2048, max_candidates=3500, max_kpc_points=32768, max_total_keypoints=1024))
        got = ctx.process_host(scans, roll=0.02, pitch=-0.015)
        k = 0
        for b in range(B):
            st = util.compare_scan(got[b], oracle.run(p, scans[b], roll=0.02, pitch=-0.015), tag=f"hdl64 {preset} {b}")
            k += st["K"]
        assert k > 0
        ctx.close()


def test_dense_128_ring_scan_radius_2m(fxlib, oracle):
    """128 x 2048, R = 2 m (BASELINE config 5): long rings, long support lists."""
    cfg = dict(n_rings=128, n_az=2048, el0_deg=-25.0, el_step_deg=40.0 / 127)
    s = capi.synth_scan(capi.synth_cfg(50, n_poles=256, **cfg))
    p = capi.params("default", n_rings=128, el0_deg=-25.0, el_step_deg=40.0 / 127, secondary_max=128, descriptor_radius=2.0)
    ctx = capi.Context(p, capi.limits(1, 128 * 2048, max_candidates=3500, max_kpc_points=65536))
    got = ctx.process_host([s], roll=0.02, pitch=-0.015)[0]
    st = util.compare_scan(got, oracle.run(p, s, roll=0.02, pitch=-0.015), tag="128 rings")
    assert st["K"] > 0
    ctx.close()


def test_feature_records_and_keypoint_records(fxlib, oracle):
    import torch
    B = 3
    scans = [util.vlp16_scan(1000 + b) for b in range(B)]
    p = capi.params("launch")
    ctx = capi.Context(p, capi.limits(B, 28800))
    got = ctx.process_host(scans, roll=0.02, pitch=-0.015)
    total = sum(g["n_keypoints"] for g in got)
    # pcl::concatenateFields -> pcl::PointDescriptor records (ref: node.cpp:119, node.h:35-53)
    buf = torch.zeros(total * capi.FX_FEATURE_RECORD_BYTES, dtype=torch.uint8, device="cuda")
    capi.check(fxlib.fx_pack_features(ctx.handle, C.c_void_p(buf.data_ptr()), total))
    ctx.synchronize()
    rec = buf.cpu().numpy().reshape(total, capi.FX_FEATURE_RECORD_BYTES)
    row = 0
    for g in got:
        for k in range(g["n_keypoints"]):
            f = rec[row].view(np.float32)
            assert f[0] == g["keypoints"][k, 0] and f[1] == g["keypoints"][k, 1] and f[2] == g["keypoints"][k, 2]
            assert f[4] == g["keypoints"][k, 3]                       # intensity @ 16
            util.assert_bit_equal(f[5:5 + 1980], g["descriptors"][k, :1980], "shape_context @ 20")
            assert (f[5 + 1980:5 + 1989] == 0).all()                  # rf @ 7940
            row += 1
    # fixed-stride keypoint records for the cross-GPU gather
    recs = torch.zeros((B, 1 + sharding.REC_KP, 4), dtype=torch.float32, device="cuda")
    ctx.pack_keypoint_records(recs.data_ptr(), sharding.REC_KP)
    ctx.synchronize()
    host = sharding.pack_records([g["keypoints"] for g in got], [g["flags"] for g in got])
    assert np.array_equal(recs.cpu().numpy().view(np.uint32), host.view(np.uint32))
    ctx.close()


def test_stage_timings_are_reported(fxlib):
    scans = [util.vlp16_scan(1000 + b) for b in range(8)]
    ctx = capi.Context(capi.params("launch"), capi.limits(8, 28800))
    ctx.set_profiling(4)
    for _ in range(5):
        ctx.process_host(scans, debug=False)
    for back in range(4):
        ms, tot = ctx.timings(back)
        assert set(ms) == set(capi.STAGE_NAMES) and all(v >= 0 for v in ms.values()) and tot > 0
    with pytest.raises(capi.FxError):
        ctx.timings(4)
    ctx.close()


def test_graph_replay_matches_plain_launches(fxlib, oracle):
    """Streaming mode: small batches replayed as one HIP graph per batch size give
    the same bits as separate launches, across changing inputs and batch sizes."""
    p = capi.params("launch")
    scans = [util.vlp16_scan(2000 + b) for b in range(6)] + [np.zeros((0, 4), np.float32)]
    plain = capi.Context(p, capi.limits(4, 28800))
    graph = capi.Context(p, capi.limits(4, 28800))
    graph.set_graph_batch(4)
    want0 = oracle.run(p, scans[0], roll=0.02, pitch=-0.015)
    for rep, part in enumerate(([0], [1], [2, 3], [6], [4, 5, 0, 6], [0], [3, 2])):
        batch = [scans[i] for i in part]
        a = plain.process_host(batch, roll=0.02, pitch=-0.015)
        g = graph.process_host(batch, roll=0.02, pitch=-0.015)
        for x, y in zip(a, g):
            assert x["n_keypoints"] == y["n_keypoints"] and x["flags"] == y["flags"]
            np.testing.assert_array_equal(x["keypoints"], y["keypoints"])
            np.testing.assert_array_equal(x["descriptors"], y["descriptors"])
            np.testing.assert_array_equal(x["filtered"], y["filtered"])
        if part[0] == 0:
            util.compare_scan(g[0], want0, tag=f"graph rep {rep}")
    plain.close()
    graph.close()
