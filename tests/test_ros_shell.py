"""f1: the ROS1 shell ros/feature_extraction_node.cpp, compiled against the stand-in headers of
tests/ros_mock (NOT roscpp: ROS cannot be installed in the development image) and driven through its two callbacks.

CPU: it compiles, and advertises / subscribes the reference's node name and topics (ref: node.cpp:41-48, 382).
GPU: a driver-style PointCloud2 (x, y, z, intensity, uint16 ring, float time; point_step 22) and an Imu go through
imuCallback / cloudCallback; the four published byte buffers are compared with the oracle and with the device-side
packers fx_pack_features / fx_pack_pointxyzi (ref: node.cpp:57-70, 72-145, 117-139)."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

from feature_extraction_amd import build, capi
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(fxlib):
    return build.build_ros_mock()


def _run(node, tmp_path, scenario_lines, params=None):
    out = tmp_path / "out"
    out.mkdir(exist_ok=True)
    (tmp_path / "scenario.txt").write_text("".join(l + "\n" for l in scenario_lines))
    env = dict(os.environ, FX_ROS_MOCK_SCENARIO=str(tmp_path / "scenario.txt"), FX_ROS_MOCK_OUT=str(out))
    for k, v in (params or {}).items():
        env["FX_ROS_PARAM_" + k] = str(v)
    r = subprocess.run([node], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return out, r.stderr


def _read_msg(out, n, topic):
    meta, fields = {}, []
    for line in (out / f"{n}_{topic}.txt").read_text().splitlines():
        k, _, v = line.partition(" ")
        if k == "field":
            name, off, dt, cnt = v.split()
            fields.append((name, int(off), int(dt), int(cnt)))
        else:
            meta[k] = v
    data = np.fromfile(out / f"{n}_{topic}.bin", dtype=np.uint8)
    return meta, fields, data


def test_shell_compiles_against_the_mock_and_keeps_the_reference_interface(fxlib, tmp_path):
    node = _build(fxlib)
    out, _ = _run(node, tmp_path, [])
    lines = (out / "node.txt").read_text().splitlines()
    assert "node feature_extraction_node" in lines  # ref: node.cpp:382
    # ref: node.cpp:41-45 publishers, :47-48 subscribers
    assert [l for l in lines if l.startswith("advertise")] == ["advertise keypoints", "advertise keypoint_cloud", "advertise cloud",
                                                                "advertise features"]
    assert sorted(l for l in lines if l.startswith("subscribe")) == ["subscribe /velodyne_points", "subscribe /xsens/data"]
    # estimate_descriptors = false: ~features is not advertised (ref: node.cpp:44-45)
    out2 = tmp_path / "b"
    out2.mkdir()
    out, _ = _run(node, out2, [], params={"estimate_descriptors": 0})
    assert "advertise features" not in (out / "node.txt").read_text()


def _quat_from_rpy(r, p, y):
    cr, sr, cp, sp, cy, sy = math.cos(r / 2), math.sin(r / 2), math.cos(p / 2), math.sin(p / 2), math.cos(y / 2), math.sin(y / 2)
    return (sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy)


def _rpy_from_quat(x, y, z, w):
    """tests/ros_mock/tf/transform_datatypes.h, restated (same double operations)."""
    s = 2.0 / (x * x + y * y + z * z + w * w)
    xs, ys, zs = x * s, y * s, z * s
    wx, wy, wz, xx, xy, xz, yy, yz, zz = w * xs, w * ys, w * zs, x * xs, x * ys, x * zs, y * ys, y * zs, z * zs
    m20, m21, m22, m10, m00 = xz - wy, yz + wx, 1.0 - (xx + yy), xy + wz, 1.0 - (yy + zz)
    pitch = -math.asin(m20)
    c = math.cos(pitch)
    return math.atan2(m21 / c, m22 / c), pitch, math.atan2(m10 / c, m00 / c)


@pytest.mark.gpu
def test_callbacks_publish_what_the_oracle_and_the_device_packers_produce(fxlib, oracle, tmp_path):
    import torch
    node = _build(fxlib)
    pts = util.vlp16_scan(1000)
    n = len(pts)
    # driver-style records: x y z intensity float32, ring uint16, time float32 -> point_step 22
    raw = np.zeros((n, 22), np.uint8)
    raw[:, 0:12] = pts[:, :3].copy().view(np.uint8).reshape(n, 12)
    raw[:, 12:16] = (np.arange(n) % 255).astype(np.float32).view(np.uint8).reshape(n, 4)
    raw[:, 16:18] = (np.arange(n) % 16).astype(np.uint16).view(np.uint8).reshape(n, 2)
    raw[:, 18:22] = np.linspace(0, 0.1, n).astype(np.float32).view(np.uint8).reshape(n, 4)
    raw.tofile(tmp_path / "cloud.bin")
    q = _quat_from_rpy(math.pi + 0.02, -0.015, 0.3)  # the sensor is mounted inverted: roll = imu_roll - pi (ref: node.cpp:65)
    imu_roll, imu_pitch, _ = _rpy_from_quat(*q)
    roll, pitch = imu_roll - math.pi, imu_pitch
    scenario = [f"imu /xsens/data {q[0]!r} {q[1]!r} {q[2]!r} {q[3]!r}",
                f"cloud /velodyne_points {tmp_path / 'cloud.bin'} velodyne 12 345 1 {n} 22 {22 * n} 0 6 "
                "x 0 7 1 y 4 7 1 z 8 7 1 intensity 12 7 1 ring 16 4 1 time 18 7 1"]
    launch = dict(cluster_tolerance=1.0, cluster_min_count=1, cluster_max_count=1000, cluster_radius_threshold=0.2,
                  number_detection_channels=2, x_max=100.0, x_min=0.0, y_max=50.0, y_min=-50.0, z_max=4.0, z_min=-1.5,
                  descriptor_radius=2.5)  # ref: launch/keypoint_playback.launch:17-33
    out, err = _run(node, tmp_path, scenario, params=launch)
    assert "capacity flags" not in err, err
    p = capi.params("launch")
    ora = oracle.run(p, pts, roll=roll, pitch=pitch)
    K = ora["n_keypoints"]
    assert K > 20
    # ---- the four messages, in the order the reference publishes them (ref: node.cpp:123, 131, 135, 139)
    order = ["features", "keypoints", "keypoint_cloud", "cloud"]
    msgs = {t: _read_msg(out, i, t) for i, t in enumerate(order)}
    for t, (meta, fields, data) in msgs.items():
        assert meta["frame_id"] == "velodyne" and meta["stamp"] == "12 345"  # header copied from the input (ref: node.cpp:121-122)
        assert meta["height"] == "1" and meta["is_bigendian"] == "0"
    xyzi_fields = [("x", 0, 7, 1), ("y", 4, 7, 1), ("z", 8, 7, 1), ("intensity", 16, 7, 1)]
    for t, key in (("keypoints", "keypoints"), ("keypoint_cloud", "kpc"), ("cloud", "filtered")):
        meta, fields, data = msgs[t]
        want = ora[key]
        assert fields == xyzi_fields and meta["point_step"] == "32" and int(meta["width"]) == len(want)
        rec = data.view(np.float32).reshape(-1, 8)
        util.assert_bit_equal(rec[:, :3], want[:, :3], f"~{t} xyz")
        util.assert_bit_equal(rec[:, 4], want[:, 3], f"~{t} intensity")
    meta, fields, data = msgs["features"]
    assert fields == xyzi_fields + [("shape_context", 20, 7, 1980), ("rf", 7940, 7, 9)]  # ref: node.h:46-53
    assert meta["point_step"] == str(capi.FX_FEATURE_RECORD_BYTES) and int(meta["width"]) == K
    rec = data.reshape(K, capi.FX_FEATURE_RECORD_BYTES)
    util.assert_bit_equal(rec[:, 0:12].copy().view(np.float32), ora["keypoints"][:, :3], "~features xyz")
    util.assert_bit_equal(rec[:, 16:20].copy().view(np.float32)[:, 0], ora["keypoints"][:, 3], "~features intensity")
    desc = rec[:, 20:20 + 4 * 1989].copy().view(np.float32)
    o = ora["descriptors"]
    assert (np.isnan(desc) == np.isnan(o)).all()
    assert np.abs(np.where(np.isnan(o), 0, desc) - np.where(np.isnan(o), 0, o)).max() <= util.DESC_TOL
    # ---- the same buffers from the device-side packers (what a zero-copy publisher would send)
    ctx = capi.Context(p, capi.limits(1, n))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_raw = torch.from_numpy(raw).cuda()
    d_xyzi = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    lay = capi.FxPc2Layout(22, 0, 4, 8, 12, 0)
    capi.check(fxlib.fx_unpack_pointcloud2(ctx.handle, C.c_void_p(d_raw.data_ptr()), n, C.byref(lay), C.c_void_p(d_xyzi.data_ptr())))
    descs = ctx.make_descs([d_xyzi.data_ptr()], [n], 16, roll, pitch)
    v = ctx.process_raw(descs, 1, capi.FX_IN_DEVICE | capi.FX_OUT_HOST)
    assert v.total_keypoints == K
    feat = torch.zeros((K * capi.FX_FEATURE_RECORD_BYTES,), dtype=torch.uint8, device="cuda")
    capi.check(fxlib.fx_pack_features(ctx.handle, C.c_void_p(feat.data_ptr()), K))
    torch.cuda.synchronize()
    dev = feat.cpu().numpy().reshape(K, capi.FX_FEATURE_RECORD_BYTES)
    # (bytes 12..15, the PCL_ADD_POINT4D pad, and the 24 tail bytes are written by both; everything must agree)
    assert np.array_equal(dev, rec), "fx_pack_features and the shell's ~features differ"
    for which, t in ((0, "keypoints"), (2, "keypoint_cloud"), (1, "cloud")):
        meta, fields, data = msgs[t]
        m = int(meta["width"])
        buf = torch.zeros((max(m, 1) * 8,), dtype=torch.float32, device="cuda")
        cnt = C.c_uint32(0)
        capi.check(fxlib.fx_pack_pointxyzi(ctx.handle, which, 0, C.c_void_p(buf.data_ptr()), max(m, 1), C.byref(cnt)))
        torch.cuda.synchronize()
        assert cnt.value == m
        assert np.array_equal(buf.cpu().numpy()[:m * 8].view(np.uint8), data), f"fx_pack_pointxyzi and the shell's ~{t} differ"
    ctx.close()
