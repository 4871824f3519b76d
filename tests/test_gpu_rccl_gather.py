"""The Python host's collective on the GPU: sharding.RcclGather (ncclAllGather on the context's own stream, what bench.py
runs at N > 1) with the ranks that are there — one on the GPU box: RCCL initialises, the gathered table equals the local
records, and two communicators on two streams do not get in each other's way."""
import numpy as np
import pytest

from feature_extraction_amd import capi, sharding
from tests import util

pytestmark = pytest.mark.gpu


def test_rccl_gather_on_context_streams(fxlib):
    import torch
    B = 4
    scans = [util.vlp16_scan(1000 + b) for b in range(B)]
    dev = torch.device("cuda", 0)
    g = sharding.RcclGather(1, 0, dev, n_comms=2)
    ctxs = [capi.Context(capi.params("launch"), capi.limits(B, 28800)) for _ in range(2)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    recs = [torch.zeros((B, 1 + sharding.REC_KP, 4), dtype=torch.float32, device=dev) for _ in range(2)]
    outs = [torch.zeros((B, 1 + sharding.REC_KP, 4), dtype=torch.float32, device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    for j, (c, st) in enumerate(zip(ctxs, streams)):
        c.set_stream(st.cuda_stream)
        c.process_host(scans, roll=0.02, pitch=-0.015, debug=False)
        c.pack_keypoint_records(recs[j].data_ptr(), sharding.REC_KP)
        g.all_gather(recs[j], outs[j], st.cuda_stream, comm=j)
    torch.cuda.synchronize()
    for j in range(2):
        assert torch.equal(outs[j], recs[j])
    per_scan = sharding.unpack_records(outs[0].cpu().numpy())
    assert sum(k for k, _, _ in per_scan) > 0 and all(f == 0 for _, f, _ in per_scan)
    g.close()
    for c in ctxs:
        c.close()
