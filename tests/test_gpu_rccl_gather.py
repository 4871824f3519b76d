"""The Python host's collective on the GPU: sharding.RcclGather (ncclAllGather / ncclGather on the context's own stream, what
bench.py runs at N > 1) with the ranks that are there — one on the GPU box: RCCL initialises, the gathered table equals the
local compact keypoint block (fx_pack_keypoint_block), the block's content equals the batch's keypoints and the host
statement of the layout (sharding.pack_block), a block too small for the batch is cut and flagged, and two communicators on
two streams do not get in each other's way."""
import numpy as np
import pytest

from feature_extraction_amd import capi, sharding
from tests import util

pytestmark = pytest.mark.gpu


def test_rccl_gather_of_compact_blocks_on_context_streams(fxlib):
    import torch
    B = 6
    scans = [util.vlp16_scan(1000 + b) for b in range(B - 1)] + [np.zeros((0, 4), np.float32)]
    dev = torch.device("cuda", 0)
    g = sharding.RcclGather(1, 0, dev, n_comms=2)
    ctxs = [capi.Context(capi.params("launch"), capi.limits(8, 28800)) for _ in range(2)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    max_scans, max_total = 8, 8 * 64  # (a block sized for more scans than the batch has: offsets beyond the batch repeat the total)
    rows = sharding.block_rows(max_scans, max_total)
    recs = [torch.full((rows, 4), 7.0, dtype=torch.float32, device=dev) for _ in range(2)]  # (stale content: the kernel rewrites every row)
    outs = [torch.zeros((rows, 4), dtype=torch.float32, device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    got = []
    for j, (c, st) in enumerate(zip(ctxs, streams)):
        c.set_stream(st.cuda_stream)
        got.append(c.process_host(scans, roll=0.02, pitch=-0.015, debug=False))
        c.pack_keypoint_block(recs[j].data_ptr(), max_scans, max_total)
        if j == 0:
            g.all_gather(recs[j], outs[j], st.cuda_stream, comm=j)
        else:
            g.gather(recs[j], outs[j], st.cuda_stream, root=0, comm=j)
    torch.cuda.synchronize()
    for j in range(2):
        assert torch.equal(outs[j], recs[j])
    blk = outs[0].cpu().numpy()
    hdr, per_scan = sharding.unpack_block(blk, max_scans)
    assert hdr == dict(scans=B, keypoints=sum(r["n_keypoints"] for r in got[0]), flags_or=0, max_total=max_total)
    for b, (n, flags, kp) in enumerate(per_scan):
        assert n == got[0][b]["n_keypoints"] and flags == got[0][b]["flags"] == 0
        assert np.array_equal(kp, got[0][b]["keypoints"][:, :4])
    want = sharding.pack_block([r["keypoints"][:, :4] for r in got[0]], [r["flags"] for r in got[0]], max_scans, max_total)
    assert np.array_equal(blk.view(np.uint32), want.view(np.uint32))
    # ---- a block too small for the batch: cut at max_total, the scans that lose keypoints (and the header) flagged
    small = 100
    rec2 = torch.zeros((sharding.block_rows(max_scans, small), 4), dtype=torch.float32, device=dev)
    ctxs[0].pack_keypoint_block(rec2.data_ptr(), max_scans, small)
    ctxs[0].synchronize()
    blk2 = rec2.cpu().numpy()
    hdr2, cut = sharding.unpack_block(blk2, max_scans)
    assert hdr2["keypoints"] == small and hdr2["flags_or"] & sharding.FX_FLAG_KP_OVERFLOW
    assert sum(n for n, _, _ in cut) == small
    for b, (n, flags, kp) in enumerate(cut):
        assert (n < got[0][b]["n_keypoints"]) == bool(flags & sharding.FX_FLAG_KP_OVERFLOW)
        assert np.array_equal(kp, got[0][b]["keypoints"][:n, :4])
    want2 = sharding.pack_block([r["keypoints"][:, :4] for r in got[0]], [r["flags"] for r in got[0]], max_scans, small)
    assert np.array_equal(blk2.view(np.uint32), want2.view(np.uint32))
    # ---- the fixed-stride records (still in the C-ABI) say the same
    rk = 256
    rec3 = torch.zeros((B, 1 + rk, 4), dtype=torch.float32, device=dev)
    ctxs[0].pack_keypoint_records(rec3.data_ptr(), rk)
    ctxs[0].synchronize()
    for b, (k, f, kp) in enumerate(sharding.unpack_records(rec3.cpu().numpy())):
        assert k == per_scan[b][0] and f == 0 and np.array_equal(kp, per_scan[b][2])
    g.close()
    for c in ctxs:
        c.close()
