"""`bench.py --gpus N` for N > 1 has never run on N devices (the SCALE run was skipped in every round; gpurun has one GPU).
This runs the script's N > 1 CONTROL FLOW with two and with three ranks on the one device there is — launched exactly as the
driver launches it (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 …`) — in its test mode
(`FX_BENCH_ONE_DEVICE=1`: gloo instead of RCCL, the gather through host memory): the launch environment, the build handshake,
per-rank seeds, agreement on the kernel to time, the barrier-bracketed regions with the MAX over the ranks and the same repeat
count everywhere, rank 0's check of the gathered table and its ONE JSON line.  Not a measurement."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world", [2, 3])
def test_bench_with_several_ranks_on_one_device(world):
    env = dict(os.environ, FX_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "6", "--warmup", "1", "--batch", "48", "--contexts", "2", "--repeats", "2",
           "--check", "2", "--no-extras", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric')]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 6 and d["repeats"] == 2 and d["scaling"] == "weak" and d["unit"] == "scans/s"
    assert d["value"] == pytest.approx(world * 48 * 6 / (d["ms_per_step"] * 6 * 1e-3), rel=1e-6)  # whole-job scans over the MAX-over-ranks time
    assert d["config"]["flags_or"] == 0 and d["config"]["gathered_record_flags_or"] == 0 and "TEST MODE" in d["config"]["parallelism"]
    assert d["parity"]["scans_checked"] == 2 and d["parity"]["keypoint_f1_vs_oracle"] == 1.0
    assert d["roofline"]["kernel"] == "k_front" and "cpu_baseline" not in d
    assert "this rank's block equals its local records" in r.stderr
