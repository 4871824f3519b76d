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


DENSE = dict(n_rings=128, n_az=2048, el0_deg=-25.0, el_step_deg=40.0 / 127)


def _dense_params(preset):
    return capi.params(preset, n_rings=128, el0_deg=-25.0, el_step_deg=40.0 / 127, secondary_max=128, descriptor_radius=2.0)


@pytest.mark.parametrize("preset", ["launch", "default"])
def test_dense_128_ring_scan_radius_2m(fxlib, oracle, preset):
    """128 x 2048, R = 2 m (BASELINE config 5): long rings, long support lists, and — under the launch preset,
    the one SURVEY.md B-6 sized the config with — about 5000 per-ring candidates a scan, which is more than one
    workgroup's LDS holds as points: the large merge tier takes those scans."""
    s = capi.synth_scan(capi.synth_cfg(50, n_poles=256, **DENSE))
    p = _dense_params(preset)
    ctx = capi.Context(p, capi.limits(1, 128 * 2048, max_candidates=8192, max_kpc_points=65536, max_keypoints=512, max_total_keypoints=512))
    got = ctx.process_host([s], roll=0.02, pitch=-0.015)[0]
    ora = oracle.run(p, s, roll=0.02, pitch=-0.015)
    st = util.compare_scan(got, ora, tag=f"128 rings {preset}")
    assert st["K"] > 0
    if preset == "launch":
        assert len(ora["candidates"]) > 4500  # really beyond the LDS tiers (<= ~3800)
    ctx.close()


def test_dense_128_ring_scan_through_the_one_small_dense_launch(fx_hooks, oracle):
    """The same scan with every dense row — support sets of up to ~10 000 points — computed by dense_slow_loop (k_desc_mid's last workgroups) instead of the
    dense tier's own kernels (what a batch gets whose predecessors had no dense row): the same result."""
    fx_hooks(FX_DENSE_SLOW=1)
    s = capi.synth_scan(capi.synth_cfg(50, n_poles=256, **DENSE))
    p = _dense_params("launch")
    ctx = capi.Context(p, capi.limits(1, 128 * 2048, max_candidates=8192, max_kpc_points=65536, max_keypoints=512, max_total_keypoints=512))
    got = ctx.process_host([s], roll=0.02, pitch=-0.015)[0]
    h = (C.c_uint32 * 8)()
    ctx.lib.fx_debug_tier_hints.argtypes = [C.c_void_p, C.c_void_p]
    capi.check(ctx.lib.fx_debug_tier_hints(ctx.handle, h))
    st = util.compare_scan(got, oracle.run(p, s, roll=0.02, pitch=-0.015), tag="128 rings, dense rows by dense_slow_loop")
    assert st["K"] > 0 and h[4] > 0  # (dense rows there were: support sets beyond the 4096-entry lists of a 262 144-point scan)
    ctx.close()


def test_128_rings_by_4096_azimuths(fxlib, oracle):
    """A sensor beyond BASELINE's largest: 128 x 4096 = 524 288 points a scan, R = 2 m, launch preset (≈ 5000 per-ring
    candidates, support sets of more than 11 000 points).  Must run unflagged and match the oracle."""
    s = capi.synth_scan(capi.synth_cfg(10, n_rings=128, n_az=4096, el0_deg=-25.0, el_step_deg=40.0 / 127, n_poles=256))
    p = _dense_params("launch")
    ctx = capi.Context(p, capi.limits(1, len(s), max_ring_points=2400, max_ring_candidates=512, max_candidates=16000, max_kpc_points=131072,
                                      max_keypoints=1024, max_total_keypoints=1024))
    got = ctx.process_host([s], roll=0.02, pitch=-0.015)[0]
    ora = oracle.run(p, s, roll=0.02, pitch=-0.015)
    st = util.compare_scan(got, ora, tag="128 x 4096")
    assert st["K"] > 100 and int(ora["kp_neighbors"].max()) > 10000
    ctx.close()


@pytest.mark.parametrize("slices", [1, 3, 8])
def test_large_merge_tier_on_vlp16_scans(fx_hooks, oracle, slices):
    """The large merge tier (cell-sorted ids + union-find in LDS, coordinates in HBM) on ordinary scans: the test
    hooks take the separate kernels (k_front has its own merge) and lower the LDS tier's capacity so that every scan with
    more than 16 candidates takes it — as ONE launch (slices 1: one workgroup a scan) and as the three launches batches of
    few scans take (k_merge_huge_a / _b / _c: 3 or 8 workgroups a scan in the pair loop, each with a union-find of its own,
    the forests united afterwards): the same components, the same keypoints."""
    fx_hooks(FX_FRONT=0, FX_MERGE_BIG_CAP=16, FX_MERGE_SLICES=slices)
    scans = [util.vlp16_scan(1000 + b) for b in range(6)] + [np.zeros((0, 4), np.float32)]
    for preset in ("launch", "default"):
        p = capi.params(preset)
        ctx = capi.Context(p, capi.limits(len(scans), 28800))
        got = ctx.process_host(scans, roll=0.02, pitch=-0.015)
        for b, sc in enumerate(scans):
            util.compare_scan(got[b], oracle.run(p, sc, roll=0.02, pitch=-0.015), tag=f"large merge tier {preset} {b}")
        ctx.close()
    # more than 192 keypoints in one scan: the order replay beyond three words of lanes (one lane, sequentially)
    crowded = [util.vlp16_scan(7 + b, n_poles=900) for b in range(2)]
    p = capi.params("launch", number_detection_channels=1)
    ctx = capi.Context(p, capi.limits(2, 28800, max_candidates=3500, max_keypoints=1024, max_total_keypoints=2048, max_kpc_points=8192))
    got = ctx.process_host(crowded, roll=0.02, pitch=-0.015)
    for b in range(2):
        st = util.compare_scan(got[b], oracle.run(p, crowded[b], roll=0.02, pitch=-0.015), tag=f"large merge tier, crowded {b}")
    assert got[0]["n_keypoints"] > 192
    ctx.close()


def _full_size(oracle, name, cfg, p, lim, B, n_uniq, sample):
    """One BASELINE configuration at its stated batch size: flags 0, batch-position independence,
    repeatability, and sampled parity against the oracle."""
    import torch
    uniq = [capi.synth_scan(capi.synth_cfg(10 + b, **cfg)) for b in range(n_uniq)]
    dev = [torch.from_numpy(s).cuda() for s in uniq]
    N = len(uniq[0])
    ctx = capi.Context(p, lim)
    descs = ctx.make_descs([dev[b % n_uniq].data_ptr() for b in range(B)], [N] * B, 16, 0.02, -0.015)
    flags = capi.FX_IN_DEVICE | capi.FX_OUT_HOST | capi.FX_OUT_CLOUDS | capi.FX_OUT_DEBUG
    res = ctx.unpack(ctx.process_raw(descs, B, flags))
    assert all(r["flags"] == 0 for r in res), [hex(r["flags"]) for r in res if r["flags"]][:4]
    for b in range(n_uniq, B):  # a scan's result does not depend on its position in the batch
        a, c = res[b], res[b - n_uniq]
        assert a["n_keypoints"] == c["n_keypoints"]
        util.assert_bit_equal(a["keypoints"], c["keypoints"], f"{name} keypoints {b}")
        util.assert_bit_equal(a["descriptors"], c["descriptors"], f"{name} descriptors {b}")
        util.assert_bit_equal(a["kpc"], c["kpc"], f"{name} keypoint_cloud {b}")
    k = 0
    for b in sample:
        st = util.compare_scan(res[b], oracle.run(p, uniq[b % n_uniq], roll=0.02, pitch=-0.015), tag=f"{name} scan {b}")
        k += st["K"]
    assert k > 0
    res2 = ctx.unpack(ctx.process_raw(descs, B, flags))
    for a, c in zip(res, res2):
        util.assert_bit_equal(a["descriptors"], c["descriptors"], f"{name} repeatability")
    ctx.close()


def test_config3_hdl64_batch_256(fxlib, oracle):
    """BASELINE config 3 at its stated size: 64 x 2048 scans, batch 256, launch preset."""
    cfg = dict(n_rings=64, n_az=2048, el0_deg=-24.8, el_step_deg=26.8 / 63, n_poles=256)
    p = capi.params("launch", n_rings=64, el0_deg=-24.8, el_step_deg=26.8 / 63, secondary_max=64)
    lim = capi.limits(256, 64 * 2048, max_candidates=4096, max_kpc_points=32768, max_keypoints=512, max_total_keypoints=256 * 256)
    _full_size(oracle, "config 3", cfg, p, lim, 256, 16, (0, 5, 11, 15, 16 + 3, 255))


def test_config5_dense_batch_64_launch_preset(fxlib, oracle):
    """BASELINE config 5 at its stated size and with the preset SURVEY.md B-6 sized it with: 128 x 2048 scans,
    R = 2 m, batch 64, launch preset (~5000 candidates and ~150 keypoints a scan, support sets beyond 5000 points)."""
    cfg = dict(n_poles=256, **DENSE)
    lim = capi.limits(64, 128 * 2048, max_candidates=8192, max_kpc_points=65536, max_keypoints=512, max_total_keypoints=64 * 256)
    _full_size(oracle, "config 5", cfg, _dense_params("launch"), lim, 64, 8, (0, 3, 7, 8 + 2, 63))


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


@pytest.mark.parametrize("front", [1, 0])
def test_stage_bytes_follow_the_batch_counts(fx_hooks, oracle, front):
    """fx_get_stage_bytes: every stage's own algorithmic bytes (the per-kernel roofline's numerator) from the batch's counts —
    with the fused front kernel (stage 0 = the scan in, every detector output out, no intermediates) and with the separate
    kernels."""
    fx_hooks(FX_FRONT=front)
    B = 3
    scans = [util.vlp16_scan(1000 + b) for b in range(B)]
    p = capi.params("launch")
    ctx = capi.Context(p, capi.limits(B, 28800))
    got = ctx.process_host(scans, roll=0.02, pitch=-0.015)
    sb = ctx.stage_bytes()
    n_f = sum(len(g["filtered"]) for g in got)
    K = sum(g["n_keypoints"] for g in got)
    assert sb["k_prep"][0] == 16.0 * 28800 * B
    prep_out = 16.0 * n_f + 28800 * B / 32.0 + 4.0 * 16 * B
    if front:
        det_out = sum(24.0 * len(g["candidates"]) + 20.0 * g["n_keypoints"] + 20.0 * len(g["kpc"]) for g in got)
        assert abs(sb["k_prep"][1] - (prep_out + det_out)) < 1e-6
        assert all(sb[k] == (0.0, 0.0) for k in ("k_bucket", "k_rings_runs", "k_rings_large", "k_merge"))
    else:
        assert abs(sb["k_prep"][1] - prep_out) < 1e-6
        assert sb["k_bucket"][0] == 16.0 * n_f and sb["k_bucket"][1] >= 16.0 * n_f  # (a window-boundary point is in two rings)
    assert sb["k_desc_group"][1] == 7956.0 * K
    support = sum(int(x) for g in got for x in g["kp_neighbors"])  # neighbours <= support points
    assert sb["k_gather"][1] >= 16.0 * support and sb["k_gather"][0] > 0
    assert all(r >= 0 and w >= 0 for r, w in sb.values())
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
