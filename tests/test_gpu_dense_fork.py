"""The dense descriptor tier beside k_desc_mid on the context's second stream (fx_api.cpp enqueue_stages: taken when the
previous batch had rows for the tier): same results as the one-stream order, on the first batch (forced through the test
build's FX_DENSE_FORK) and on later batches of the product library (by the hint), also when a graph replays the batch."""
import numpy as np
import pytest

from feature_extraction_amd import capi
from tests import util
from tests.test_gpu_fuzz_dense import dense_case

pytestmark = pytest.mark.gpu

DENSE = dict(n_rings=128, n_az=2048, el0_deg=-25.0, el_step_deg=40.0 / 127)


def _params():
    return capi.params("launch", n_rings=128, el0_deg=-25.0, el_step_deg=40.0 / 127, secondary_max=128, descriptor_radius=2.0)


@pytest.mark.parametrize("fork", [1, 0])
def test_forced_fork_dense_128_ring_scan(fx_hooks, oracle, fork):
    fx_hooks(FX_DENSE_FORK=fork)
    s = capi.synth_scan(capi.synth_cfg(50, n_poles=256, **DENSE))
    p = _params()
    ctx = capi.Context(p, capi.limits(1, 128 * 2048, max_candidates=8192, max_kpc_points=65536, max_keypoints=512, max_total_keypoints=512))
    ora = oracle.run(p, s, roll=0.02, pitch=-0.015)
    for rep in range(3):  # (plain launches, then the batch captured as a graph, then the graph replayed)
        if rep == 1:
            ctx.set_graph_batch(1)
        got = ctx.process_host([s], roll=0.02, pitch=-0.015)[0]
        st = util.compare_scan(got, ora, tag=f"fork {fork} call {rep}")
    assert st["K"] > 0 and int(np.max(got["kp_neighbors"])) > 1024  # rows of the dense tier
    ctx.close()


def test_fork_by_the_hint_on_later_batches(fxlib, oracle):
    """Product library: the first batch knows nothing (one stream), the following ones fork; two different scans take turns."""
    scans = [capi.synth_scan(capi.synth_cfg(60 + i, n_poles=200, **DENSE)) for i in range(2)]
    p = _params()
    ctx = capi.Context(p, capi.limits(8, 128 * 2048, max_candidates=8192, max_kpc_points=65536, max_keypoints=512, max_total_keypoints=8 * 256))
    ora = [oracle.run(p, s, roll=0.02, pitch=-0.015) for s in scans]
    for rep in range(3):
        batch = [scans[(rep + b) % 2] for b in range(8)]
        got = ctx.process_host(batch, roll=0.02, pitch=-0.015)
        for b in (0, 1, 7):
            util.compare_scan(got[b], ora[(rep + b) % 2], tag=f"batch {rep} scan {b}")
    ctx.close()


def test_forced_fork_dense_fuzz_seeds(fx_hooks, oracle):
    fx_hooks(FX_DENSE_FORK=1)
    checked = 0
    for seed in range(7100, 7116):
        s, p, roll, pitch, lim, what = dense_case(seed)
        ctx = capi.Context(p, lim)
        got = ctx.process_host([s], roll=roll, pitch=pitch)[0]
        ctx.close()
        if got["flags"]:
            continue
        util.compare_scan(got, oracle.run(p, s, roll=roll, pitch=pitch), tag=f"dense seed {seed} {what}")
        checked += 1
    assert checked >= 12
