"""f4: the streaming front end fx::StreamBatcher (csrc/fx_batcher.hpp).  Four simulated 10 Hz sensors
(one producer thread each) push scans; the consumer batches whatever has arrived; every scan's keypoints and descriptors
must be what the oracle gives for that scan alone, whichever batch it rode in.  Also a burst (all scans at once), where
the batches have to grow.  Prints the latency distribution the driver measured."""
import re
import struct
import subprocess

import numpy as np
import pytest

from feature_extraction_amd import build, capi
from tests import util

pytestmark = pytest.mark.gpu


def _read(path):
    raw = open(path, "rb").read()
    pos, out = 0, {}
    while pos < len(raw):
        sensor, seq, flags, K = struct.unpack_from("<4I", raw, pos)
        pos += 16
        kp = np.frombuffer(raw, np.float32, K * 4, pos).reshape(K, 4)
        pos += K * 16
        desc = np.frombuffer(raw, np.float32, K * capi.FX_DESC_FLOATS, pos).reshape(K, capi.FX_DESC_FLOATS)
        pos += K * capi.FX_DESC_FLOATS * 4
        out[(sensor, seq)] = (flags, kp, desc)
    return out


def _check(got, oracle, sensors, per_sensor, poles=None, min_k=0):
    p = capi.params("launch")
    assert len(got) == sensors * per_sensor
    total_k = 0
    for (s, q), (flags, kp, desc) in sorted(got.items()):
        scan = util.vlp16_scan(1000 + 1000 * s + q) if poles is None else capi.synth_scan(capi.synth_cfg(1000 + 1000 * s + q, n_poles=poles))
        ora = oracle.run(p, scan, roll=0.02, pitch=-0.015)
        assert len(ora["keypoints"]) >= min_k
        assert flags == 0
        util.assert_bit_equal(kp, ora["keypoints"], f"sensor {s} scan {q} keypoints")
        o = ora["descriptors"]
        assert desc.shape == o.shape and (np.isnan(desc) == np.isnan(o)).all()
        assert np.abs(np.where(np.isnan(o), 0, desc) - np.where(np.isnan(o), 0, o)).max(initial=0.0) <= util.DESC_TOL
        total_k += len(kp)
    assert total_k > 0


def test_four_paced_sensors_match_the_oracle_scan_by_scan(fxlib, oracle, tmp_path):
    exe = build.build_batcher()
    out = tmp_path / "paced.bin"
    r = subprocess.run([exe, "--sensors", "4", "--hz", "10", "--seconds", "1.5", "--out", str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    print(r.stdout.strip())
    m = re.search(r"p50 ([\d.]+) p90 ([\d.]+) p99 ([\d.]+)", r.stdout)
    assert m and float(m.group(3)) < 100.0, r.stdout  # a 10 Hz sensor's period: a scan never waits for the next one
    _check(_read(out), oracle, 4, 15)


def test_burst_grows_the_batches(fxlib, oracle, tmp_path):
    exe = build.build_batcher()
    out = tmp_path / "burst.bin"
    r = subprocess.run([exe, "--sensors", "4", "--burst", "12", "--out", str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    print(r.stdout.strip())
    m = re.search(r"(\d+) scans in (\d+) batches \(largest (\d+)\)", r.stdout)
    assert m and int(m.group(1)) == 48 and int(m.group(2)) < 48 and int(m.group(3)) > 1, r.stdout
    _check(_read(out), oracle, 4, 12)


def test_scans_with_more_keypoints_than_the_pool_average(fxlib, oracle, tmp_path):
    """a one- or two-scan batch with more than 64 keypoints a scan (the default pool's average) must not read past
    the pool: the batcher sizes the pool for max_batch * max_keypoints rows and clamps what it copies."""
    exe = build.build_batcher()
    out = tmp_path / "poles.bin"
    r = subprocess.run([exe, "--sensors", "1", "--burst", "3", "--poles", "256", "--max-batch", "2", "--out", str(out)], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    _check(_read(out), oracle, 1, 3, poles=256, min_k=65)
