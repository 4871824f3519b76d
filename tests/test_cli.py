"""The C++ host front end (csrc/fx_node.hpp + csrc/fx_cli.cpp): BASELINE config 1, one sweep from
a .pcd file — without ROS."""
import os
import subprocess

import numpy as np
import pytest

from feature_extraction_amd import build, capi
from tests import util


def _read_pcd(path):
    raw = open(path, "rb").read()
    head, _, body = raw.partition(b"DATA binary\n")
    n = int([l for l in head.decode().splitlines() if l.startswith("POINTS")][0].split()[1])
    return np.frombuffer(body, np.float32, n * 4).reshape(n, 4)


@pytest.fixture(scope="module")
def cli(fxlib):
    return build.build_cli()


def test_synth_pcd_matches_generator(cli, tmp_path):
    pcd = str(tmp_path / "s.pcd")
    subprocess.check_call([cli, "--synth", "1000", pcd])
    assert np.array_equal(_read_pcd(pcd), util.vlp16_scan(1000))


def test_cli_refuses_without_gpu(cli, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    pcd = str(tmp_path / "s.pcd")
    subprocess.check_call([cli, "--synth", "1", pcd])
    r = subprocess.run([cli, pcd], capture_output=True, text=True)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_cli_pcd_sweep_matches_oracle(cli, oracle, tmp_path):
    pcd, out = str(tmp_path / "s.pcd"), str(tmp_path / "o")
    subprocess.check_call([cli, "--synth", "1000", pcd])
    r = subprocess.run([cli, pcd, "--launch", "--roll", "0.02", "--pitch", "-0.015", "--out", out], capture_output=True,
                       text=True, check=True)
    p = capi.params("launch")
    ora = oracle.run(p, util.vlp16_scan(1000), roll=0.02, pitch=-0.015)
    assert f"keypoints {ora['n_keypoints']} " in r.stdout and "flags 0x0" in r.stdout
    util.assert_bit_equal(_read_pcd(out + "_keypoints.pcd"), ora["keypoints"], "~keypoints")
    util.assert_bit_equal(_read_pcd(out + "_cloud.pcd"), ora["filtered"], "~cloud")
    util.assert_bit_equal(_read_pcd(out + "_keypoint_cloud.pcd"), ora["kpc"], "~keypoint_cloud")
    d = np.fromfile(out + "_descriptors.f32", np.float32).reshape(-1, 1989)
    assert np.abs(d - ora["descriptors"]).max() <= util.DESC_TOL
