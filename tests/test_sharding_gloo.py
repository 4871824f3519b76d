"""N > 1 path on CPU: two gloo ranks shard a scan stream, each builds the keypoint records of its
block, one all-gather assembles the table; rank 0 checks it against the unsharded result.
(The record producer here is the oracle — the GPU producer is checked in test_gpu_*.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from feature_extraction_amd import capi, sharding
from oracle import oracle_py
from tests import util

TOTAL = 6
REC_KP = 256  # bench.py's record stride: the contexts' keypoint capacity (limits.max_keypoints), never truncating
_CACHE = {}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _records_for(span, rec_kp=REC_KP):
    p = capi.params("launch")
    kps, flags = [], []
    for b in range(*span):
        if b not in _CACHE:
            _CACHE[b] = oracle_py.run(p, util.vlp16_scan(1000 + b, n_az=450), roll=0.02, pitch=-0.015)["keypoints"]
        kps.append(_CACHE[b])
        flags.append(0)
    return sharding.pack_records(kps, flags, rec_kp=rec_kp) if kps else np.zeros((0, 1 + rec_kp, 4), np.float32)


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    span = sharding.shard_range(TOTAL, world, rank)
    rec = torch.from_numpy(_records_for(span))
    gathered = sharding.all_gather_records(rec, world)
    # the form bench.py uses: preallocated table, collective left in flight, waited for later
    table = torch.zeros((TOTAL, 1 + REC_KP, 4), dtype=torch.float32)
    same, work = sharding.all_gather_records(rec, world, out=table, async_op=True)
    work.wait()
    assert same is table and torch.equal(table, gathered)
    # max-over-ranks timing plumbing used by bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        np.save(out_path, gathered.numpy())
        assert t.item() == float(world)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_equals_unsharded(tmp_path):
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    want = _records_for((0, TOTAL))
    assert got.shape == want.shape == (TOTAL, 1 + REC_KP, 4)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    per_scan = sharding.unpack_records(got)
    assert sum(k for k, _, _ in per_scan) > 0 and all(f == 0 for _, f, _ in per_scan)


def _worker_uneven(rank, world, port, total, rec_kp, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    span = sharding.shard_range(total, world, rank)
    rec = torch.from_numpy(sharding.pad_block(_records_for(span, rec_kp), total, world))  # short blocks padded with empty records
    assert rec.shape[0] == sharding.block_size(total, world)
    table = sharding.all_gather_records(rec, world)
    if rank == 0:
        np.save(out_path, table.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_a_stream_that_does_not_divide_and_a_record_stride_of_512(tmp_path):
    """the node's shape — eight ranks —, a stream of 21 scans (blocks of two and three: every rank hands the
    collective three records, the short blocks padded) and records of 512 keypoints."""
    total, world, rec_kp = 21, 8, 512
    out = str(tmp_path / "gathered8.npy")
    mp.spawn(_worker_uneven, args=(world, _free_port(), total, rec_kp, out), nprocs=world, join=True)
    table = np.load(out)
    assert table.shape == (world * sharding.block_size(total, world), 1 + rec_kp, 4)
    got = sharding.stream_order(table, total, world)
    want = _records_for((0, total), rec_kp)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert sum(k for k, _, _ in sharding.unpack_records(got)) > 0


# ---- the compact keypoint block (fx_pack_keypoint_block: what crosses GPUs since 0.7) ------------------------------------------

def _block_for(span, max_scans, max_total):
    p = capi.params("launch")
    kps = []
    for b in range(*span):
        if b not in _CACHE:
            _CACHE[b] = oracle_py.run(p, util.vlp16_scan(1000 + b, n_az=450), roll=0.02, pitch=-0.015)["keypoints"]
        kps.append(_CACHE[b])
    return sharding.pack_block(kps, [0] * len(kps), max_scans, max_total), kps


def _worker_blocks(rank, world, port, total, max_total, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    max_scans = sharding.block_size(total, world)
    blk, _ = _block_for(sharding.shard_range(total, world, rank), max_scans, max_total)
    rec = torch.from_numpy(blk)
    table = sharding.all_gather_records(rec, world)                 # on every rank (ncclAllGather's shape)
    rooted = sharding.gather_records_to_root(rec, world, root=0)    # on rank 0 only (ncclGather's shape)
    assert (rooted is None) == (rank != 0)
    if rank == 0:
        assert torch.equal(rooted, table)
        np.save(out_path, table.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total,world,max_total", [(6, 2, 3 * 256), (21, 8, 3 * 256), (21, 8, 40)])
def test_compact_blocks_gathered_equal_the_unsharded_stream(tmp_path, total, world, max_total):
    """World 2 and the node's shape, world 8 with a stream that does not divide (blocks of two and three scans): every rank
    hands the collective ONE fixed-size block — offsets, flags, its keypoints packed in scan order —, the table is gathered on
    every rank (all-gather) and on rank 0 only (gather); the stream comes back in order.  With a block too small for a rank's
    keypoints (max_total 40) the scans that lose keypoints are flagged, the others intact."""
    out = str(tmp_path / "blocks.npy")
    mp.spawn(_worker_blocks, args=(world, _free_port(), total, max_total, out), nprocs=world, join=True)
    table = np.load(out)
    max_scans = sharding.block_size(total, world)
    assert table.shape == (world * sharding.block_rows(max_scans, max_total), 4)
    got = sharding.stream_order_blocks(table, total, world, max_scans)
    _, want = _block_for((0, total), total, 1 << 20)
    assert len(got) == total and sum(n for n, _, _ in got) > 0
    cut = 0
    for s, (n, flags, kp) in enumerate(got):
        full = len(want[s])
        assert np.array_equal(kp, np.asarray(want[s], np.float32)[:n])
        if n < full:
            cut += 1
            assert flags & sharding.FX_FLAG_KP_OVERFLOW
    assert (cut > 0) == (max_total == 40)
    # bytes at bench.py's shape (1024 scans a rank, 64 keypoints a scan of block capacity) against the fixed-stride records it
    # gathered until 0.6 (stride 256): a quarter
    assert sharding.block_rows(1024, 64 * 1024) / (1024 * (1 + 256)) < 0.26
