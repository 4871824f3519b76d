"""PointCloud2 wire formats: ingress by field offsets out of driver-style records,
egress as the 32-byte PointXYZI records pcl_ros publishes (ref: node.cpp:79-81, 129-139)."""
import ctypes as C

import numpy as np
import pytest

from feature_extraction_amd import capi
from tests import util

pytestmark = pytest.mark.gpu


def _velodyne_msg(pts, point_step, offs, big_endian=False):
    """Driver-style records: x, y, z, intensity float32 at the given offsets, uint16 ring after them, junk elsewhere."""
    n = len(pts)
    raw = np.random.default_rng(1).integers(0, 255, (n, point_step), dtype=np.uint8)
    for col, off in enumerate(offs):
        v = pts[:, col].astype(">f4" if big_endian else "<f4").view(np.uint8).reshape(n, 4)
        raw[:, off:off + 4] = v
    return raw


@pytest.mark.parametrize("point_step,offs,big", [(22, (0, 4, 8, 16), False), (32, (0, 4, 8, 16), False),
                                                 (19, (3, 7, 11, 15), False), (48, (16, 20, 24, 40), True)])
def test_ingress_then_pipeline_matches_oracle(fxlib, oracle, point_step, offs, big):
    import torch
    pts = util.vlp16_scan(1000)
    pts[:, 3] = np.arange(len(pts)) % 255  # incoming intensity: carried over by the unpack, ignored by the path
    raw = _velodyne_msg(pts, point_step, offs, big)
    d_raw = torch.from_numpy(raw).cuda()
    d_xyzi = torch.zeros((len(pts), 4), dtype=torch.float32, device="cuda")
    p = capi.params("launch")
    ctx = capi.Context(p, capi.limits(1, 28800))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    lay = capi.FxPc2Layout(point_step, offs[0], offs[1], offs[2], offs[3], 1 if big else 0)
    capi.check(fxlib.fx_unpack_pointcloud2(ctx.handle, C.c_void_p(d_raw.data_ptr()), len(pts), C.byref(lay),
                                           C.c_void_p(d_xyzi.data_ptr())))
    torch.cuda.synchronize()
    assert np.array_equal(d_xyzi.cpu().numpy().view(np.uint32), pts.view(np.uint32))
    descs = ctx.make_descs([d_xyzi.data_ptr()], [len(pts)], 16, 0.02, -0.015)
    flags = capi.FX_IN_DEVICE | capi.FX_OUT_HOST | capi.FX_OUT_CLOUDS | capi.FX_OUT_DEBUG
    got = ctx.unpack(ctx.process_raw(descs, 1, flags))[0]
    ora = oracle.run(p, pts, roll=0.02, pitch=-0.015)
    util.compare_scan(got, ora, tag=f"pc2 step {point_step}")
    # egress: the three PointXYZI topics as 32-byte records
    for which, key in ((0, "keypoints"), (1, "filtered"), (2, "kpc")):
        want = ora[key]
        buf = torch.full((max(len(want), 1) * 8,), -7.0, dtype=torch.float32, device="cuda")
        n = C.c_uint32(0)
        capi.check(fxlib.fx_pack_pointxyzi(ctx.handle, which, 0, C.c_void_p(buf.data_ptr()), max(len(want), 1), C.byref(n)))
        torch.cuda.synchronize()
        assert n.value == len(want)
        rec = buf.cpu().numpy().reshape(-1, 8)[:n.value]
        util.assert_bit_equal(rec[:, :3], want[:, :3], f"{key} xyz @0")
        util.assert_bit_equal(rec[:, 4], want[:, 3], f"{key} intensity @16")
        assert (rec[:, 3] == 1.0).all() and (rec[:, 5:] == 0).all()
    ctx.close()


def test_layout_errors(fxlib):
    ctx = capi.Context(capi.params("default"), capi.limits(1, 1024))
    lay = capi.FxPc2Layout(16, 0, 4, 14, 0xffffffff, 0)
    assert fxlib.fx_unpack_pointcloud2(ctx.handle, C.c_void_p(16), 4, C.byref(lay), C.c_void_p(16)) == 1
    n = C.c_uint32(0)
    assert fxlib.fx_pack_pointxyzi(ctx.handle, 0, 5, C.c_void_p(16), 4, C.byref(n)) == 1
    ctx.close()
