"""C++ multi-GPU driver (csrc/fx_multi.hpp) on the devices that are there: RCCL communicators come up, every rank runs
its block of the batch, ncclAllGather assembles the keypoint table, and the table equals the producing ranks' results.
The GPU box has one device, so this is the nranks = 1 run SURVEY.md 8e asks the RCCL path to be testable with."""
import subprocess

import pytest

from feature_extraction_amd import build

pytestmark = pytest.mark.gpu


def test_fx_multi_cli_runs_the_rccl_gather():
    exe = build.build_multi()
    r = subprocess.run([exe, "--batch", "64", "--steps", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "RCCL communicators up" in r.stdout and "gathered table == per-rank results" in r.stdout, r.stdout
