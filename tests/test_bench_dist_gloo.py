"""bench.py's distributed scaffolding (feature_extraction_amd/bench_dist.py) end to end on two CPU ranks (gloo) with a stub step
— everything `bench.py --gpus N` does around the GPU work, which no multi-GPU box has run yet: local rank 0
builds while the other polls, the ranks agree on an optional facility, rank 0's dominant kernel is the one every rank times,
the timed region is barrier-bracketed with the MAX over the ranks and repeated the same number of times everywhere, the
seeds of the ranks' batches are disjoint, the step loop keeps one collective in flight per slot and drains them."""
import os
import socket
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from feature_extraction_amd import bench_dist, sharding

B, REC_KP, K, STEPS = 5, 8, 3, 7


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    lib = os.path.join(tmp, "lib.so")
    built = []

    def build():  # (local rank 0: a slow build, linked to a temporary name and renamed like build.py's)
        time.sleep(1.0)
        with open(lib + ".tmp", "w") as f:
            f.write("x")
        os.replace(lib + ".tmp", lib)
        built.append(1)

    waited = bench_dist.wait_for_library(rank, build, lambda: not os.path.exists(lib), timeout_s=60, poll_s=0.05)
    assert os.path.exists(lib) and ((rank == 0 and built) or (rank > 0 and not built and waited > 0.3))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    coll = bench_dist.Collectives(torch, dist, "cpu", rank, world, True)
    # an optional facility is used by all ranks or by none
    assert coll.agree(True) is True and coll.agree(rank == 0) is False and coll.agree(rank != 0) is False
    # every rank times the kernel rank 0 chose
    assert coll.broadcast_index(3 + rank) == 3
    assert coll.max(rank + 1.0) == float(world) and coll.sum(rank + 1.0) == world * (world + 1) / 2
    # seeds: a rank's two alternating batches and every other rank's are pairwise disjoint
    mine = [bench_dist.rank_seeds(rank, world, B), bench_dist.rank_seeds(rank, world, B, second=True)]
    every = [None] * world
    dist.all_gather_object(every, mine)
    flat = [s for r in every for batch in r for s in batch]
    assert len(flat) == len(set(flat)) == 2 * world * B and min(flat) == 1000

    # ---- the step loop with a stub step: slot j writes (rank, step number, which batch) records and leaves their all-gather in flight
    recs = [torch.zeros((B, 1 + REC_KP, 4), dtype=torch.float32) for _ in range(K)]
    gathered = [torch.zeros((world * B, 1 + REC_KP, 4), dtype=torch.float32) for _ in range(K)]
    log = []

    def run_slot(j, which):
        n = len(log)
        log.append((j, which))
        time.sleep(0.002 * (rank + 1))  # (ranks of different speed: the MAX over the ranks is the slow one's)
        recs[j][:] = 0
        recs[j][:, 0, 0] = rank
        recs[j][:, 0, 1] = n
        recs[j][:, 0, 2] = which
        return sharding.all_gather_records(recs[j], world, out=gathered[j], async_op=True)[1]

    loop = bench_dist.StepLoop(K, run_slot)
    regions = bench_dist.measure(STEPS, loop.step, loop.drain, lambda: None, coll, repeats=0, target_seconds=0.25)
    assert all(p is None for p in loop.pending)
    # the same number of regions on every rank (or one of them would wait in a barrier for ever), the same times (MAX)
    every = [None] * world
    dist.all_gather_object(every, regions)
    assert every[0] == every[1] and len(regions) >= 2 and len(log) == STEPS * len(regions)
    assert min(regions) >= 0.002 * world * STEPS * 0.9  # (the slow rank's time)
    # slots in turn, the two batches alternating every K steps
    assert log[:2 * K + 1] == [(i % K, (i // K) % 2) for i in range(2 * K + 1)]
    # the last collective of every slot: each rank's block carries that rank's id and the same step number
    for j in range(K):
        g = gathered[j].view(world, B, 1 + REC_KP, 4)
        assert [int(g[r, 0, 0, 0]) for r in range(world)] == list(range(world))
        assert len({int(g[r, 0, 0, 1]) for r in range(world)}) == 1
    assert loop.last_slot == (len(log) - 1) % K
    # an explicit repeat count wins
    assert len(bench_dist.measure(2, loop.step, loop.drain, lambda: None, coll, repeats=3)) == 3
    dist.barrier()
    dist.destroy_process_group()


def test_bench_scaffolding_on_two_gloo_ranks(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)


def test_single_process_is_the_identity():
    coll = bench_dist.Collectives(torch, dist, "cpu", 0, 1, False)
    assert coll.agree(True) and not coll.agree(False) and coll.max(2.5) == 2.5 and coll.sum(4) == 4.0 and coll.broadcast_index(7) == 7
    coll.barrier()
    assert bench_dist.repeat_count(0.01, 0, 1.0) == 100 and bench_dist.repeat_count(0.01, 5, 1.0) == 5 and bench_dist.repeat_count(10.0, 0, 1.0) == 1
    assert bench_dist.repeat_count(1e-6, 0, 1.0) == 200
    calls = []
    loop = bench_dist.StepLoop(2, lambda j, which: calls.append((j, which)))
    regions = bench_dist.measure(4, loop.step, loop.drain, lambda: None, coll, repeats=2)
    assert len(regions) == 2 and calls == [(0, 0), (1, 0), (0, 1), (1, 1)] * 2
    assert np.isfinite(regions).all()
