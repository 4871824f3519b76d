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


@pytest.mark.parametrize("ranks,batch,inflight", [(8, 61, 3), (2, 64, 2), (3, 2, 1)])
def test_many_ranks_on_one_device_through_the_host_gather(ranks, batch, inflight):
    """fx::MultiGpu has never run on more than one device.  Its self-test mode puts G worker threads — tickets,
    the error barrier before the collective, slots in flight, uneven blocks (61 scans over 8 ranks; 2 scans over 3: a rank
    with none) — on the ONE device of this box, the collective replaced by a gather through host memory (RCCL refuses a
    communicator with the same device twice).  The gathered table must equal every producing rank's results, every rank
    must hold the same table, a batch that one rank cannot run must fail on all of them, and the next one be right."""
    exe = build.build_multi()
    r = subprocess.run([exe, "--selftest", str(ranks), "--batch", str(batch), "--steps", "6", "--inflight", str(inflight), "--bad-scan", str(batch - 1)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"{ranks} rank(s)" in r.stdout and "SELF-TEST" in r.stdout and "gathered table == per-rank results" in r.stdout, r.stdout
    assert "failed on every rank as it must" in r.stdout and "the batch after the failed one equals the reference table" in r.stdout, r.stdout
