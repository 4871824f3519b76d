"""The C++ host's sharding plan and record layout (csrc/fx_shard.hpp — what csrc/fx_multi.hpp shards and gathers with)
on CPU: one process per rank packs the records of its block of a stream; concatenated in rank order they must equal the
unsharded table, and the Python plan (feature_extraction_amd/sharding.py, which bench.py and the gloo test use) must
agree with the C++ one.  The RCCL collective itself runs in tests/test_gpu_multi.py (one rank on the 1-GPU box)."""
import struct
import subprocess

import numpy as np
import pytest

from feature_extraction_amd import build, sharding


@pytest.mark.parametrize("total,world,rec_kp", [(6, 2, 127), (7, 2, 256), (5, 3, 256), (3, 4, 64), (21, 8, 512), (8, 8, 512), (5, 8, 256)])
def test_ranks_tile_the_stream_and_records_match(tmp_path, total, world, rec_kp):
    build.build_multi()
    exe = build.SELFTEST
    rng = np.random.default_rng(total * 10 + world)
    kps = [rng.normal(size=(int(rng.integers(0, 140)), 4)).astype(np.float32) for _ in range(total)]  # some beyond 127
    src = tmp_path / "in.bin"
    with open(src, "wb") as f:
        for kp in kps:
            f.write(struct.pack("<I", len(kp)))
            f.write(kp.tobytes())
    procs = []
    for r in range(world):  # one process per rank, as on a node with `world` GPUs
        procs.append(subprocess.Popen([exe, str(total), str(world), str(r), str(src), str(tmp_path / f"out{r}.bin"), str(rec_kp)],
                                      stdout=subprocess.PIPE, text=True))
    spans = []
    for r, pr in enumerate(procs):
        out, _ = pr.communicate(timeout=60)
        assert pr.returncode == 0, (r, pr.returncode)
        spans.append(tuple(int(x) for x in out.split()))
    assert spans == [sharding.shard_range(total, world, r) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == total and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    # every rank hands the collective block_size records (short blocks padded with empty ones): the gathered table
    bs = sharding.block_size(total, world)
    table = np.concatenate([np.fromfile(tmp_path / f"out{r}.bin", np.float32) for r in range(world)]).reshape(world * bs, 1 + rec_kp, 4)
    want = sharding.pack_records(kps, [0] * total, rec_kp=rec_kp)  # (a stride below a scan's count truncates and flags it)
    got = sharding.stream_order(table, total, world)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    for s in range(total):
        o = sharding.owner_of(s, total, world)
        assert spans[o][0] <= s < spans[o][1]


@pytest.mark.parametrize("total,world,max_total", [(6, 2, 400), (7, 2, 100), (5, 3, 200), (21, 8, 256), (5, 8, 64), (8, 8, 20)])
def test_compact_keypoint_blocks_tile_the_stream(tmp_path, total, world, max_total):
    """The compact keypoint block (fx_pack_keypoint_block's layout; what fx::MultiGpu and bench.py gather since 0.7): one process
    per rank packs the block of its share with the C++ statement (fx::pack_block); the blocks in rank order are what an all-gather
    assembles; the Python statement (sharding.pack_block) must give the same bytes and the stream must come back in order — cut,
    and flagged, where a rank's keypoints exceed the block (max_total below some shares' totals on purpose)."""
    build.build_multi()
    rng = np.random.default_rng(total * 100 + world)
    kps = [rng.normal(size=(int(rng.integers(0, 90)), 4)).astype(np.float32) for _ in range(total)]
    src = tmp_path / "in.bin"
    with open(src, "wb") as f:
        for kp in kps:
            f.write(struct.pack("<I", len(kp)))
            f.write(kp.tobytes())
    max_scans = sharding.block_size(total, world)
    blocks = []
    for r in range(world):
        out = tmp_path / f"blk{r}.bin"
        pr = subprocess.run([build.SELFTEST, str(total), str(world), str(r), str(src), str(out), "block", str(max_total)], capture_output=True, text=True, timeout=60)
        assert pr.returncode == 0, (r, pr.returncode)
        blk = np.fromfile(out, np.float32).reshape(-1, 4)
        lo, hi = sharding.shard_range(total, world, r)
        want = sharding.pack_block(kps[lo:hi], [0] * (hi - lo), max_scans, max_total)
        assert blk.shape == want.shape == (sharding.block_rows(max_scans, max_total), 4)
        assert np.array_equal(blk.view(np.uint32), want.view(np.uint32)), r
        blocks.append(blk)
    got = sharding.stream_order_blocks(np.concatenate(blocks), total, world, max_scans)
    assert len(got) == total
    cut = 0
    for s, (n, flags, kp) in enumerate(got):
        assert n <= len(kps[s]) and np.array_equal(kp, kps[s][:n])
        assert (n < len(kps[s])) == bool(flags & sharding.FX_FLAG_KP_OVERFLOW)
        cut += n < len(kps[s])
    if max_total >= 400:
        assert cut == 0
