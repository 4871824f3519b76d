"""The distributed scaffolding of bench.py, separate from the GPU work so that it runs end to end on CPU ranks (gloo, world 2:
tests/test_bench_dist_gloo.py) with a stub step: who builds the library and who waits, agreement on an optional facility,
every rank timing the same kernel, the timed region (barrier + synchronize on both sides, MAX over the ranks), the repeat count
every rank derives from the first region, per-rank seeds, and the step loop with its collectives left in flight.

Nothing here touches a device: `dev` is whatever device the collectives' tensors live on ("cpu" under gloo), `sync` is the
caller's device synchronisation (a no-op on CPU)."""
import time

import numpy as np


def wait_for_library(local_rank, build, stale, timeout_s=900.0, poll_s=0.5, log=None):
    """Local rank 0 builds (a stale library would otherwise be rewritten by every rank at once); the others poll until the
    library is there and current (build.py links to a temporary name and renames: a library that is there is whole).
    Raises SystemExit when the wait runs out."""
    if local_rank == 0:
        build()
        return 0.0
    t0 = time.time()
    while stale():
        if time.time() - t0 > timeout_s:
            raise SystemExit(f"[bench] local rank {local_rank}: the library is still stale after {timeout_s:.0f} s (did local rank 0's build fail?)")
        time.sleep(poll_s)
    waited = time.time() - t0
    if log and waited > 1.0:
        log(f"[bench] local rank {local_rank} waited {waited:.1f} s for local rank 0's build")
    return waited


def rank_seeds(rank, world, batch, second=False, base=1000):
    """Seeds of the synthetic scans a rank owns: base + global scan index; `second`: the other batch the steps alternate
    with (no context ever sees the batch it processed last time) — disjoint from every rank's first batch."""
    off = (world + rank) if second else rank
    return [base + off * batch + b for b in range(batch)]


class Collectives:
    """The few collectives the harness needs, over torch.distributed when the job has more than one rank (or when forced),
    as the identity otherwise."""

    def __init__(self, torch, dist, dev, rank, world, active):
        self.torch, self.dist, self.dev, self.rank, self.world = torch, dist, dev, rank, world
        self.active = bool(active)  # a process group exists

    def _reduce(self, value, dtype, op):
        if not self.active or self.world == 1:
            return value
        t = self.torch.tensor([value], dtype=dtype, device=self.dev)
        self.dist.all_reduce(t, op=op)
        return t.item()

    def agree(self, ok):
        """True only when EVERY rank says so (an optional facility — RCCL's C API — is used by all ranks or by none)."""
        return bool(self._reduce(1 if ok else 0, self.torch.int64, self.dist.ReduceOp.MIN if self.active else None))

    def max(self, x):
        return float(self._reduce(float(x), self.torch.float64, self.dist.ReduceOp.MAX if self.active else None))

    def sum(self, x):
        return float(self._reduce(float(x), self.torch.float64, self.dist.ReduceOp.SUM if self.active else None))

    def broadcast_index(self, idx, src=0):
        """Rank `src`'s choice, on every rank (every rank times the same kernel)."""
        if not self.active:
            return idx
        t = self.torch.tensor([idx], dtype=self.torch.int64, device=self.dev)
        self.dist.broadcast(t, src)
        return int(t.item())

    def barrier(self):
        if self.active and self.world > 1:
            self.dist.barrier()


def timed_region(steps, step, drain, sync, coll):
    """EXACTLY `steps` steps between barrier + synchronize on both sides; returns the MAX over the ranks (seconds)."""
    coll.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    drain()
    sync()
    coll.barrier()
    return coll.max(time.perf_counter() - t0)


def repeat_count(first_region_s, repeats, target_seconds, cap=200):
    """Timed regions to run: `repeats` when given, else as many as fill `target_seconds` — derived from the first region's
    all-reduced time, so every rank arrives at the same count (a rank that ran one region more would wait in its barrier
    for ever)."""
    if repeats > 0:
        return repeats
    return max(1, min(cap, int(np.ceil(target_seconds / first_region_s)))) if first_region_s > 0 else 1


def measure(steps, step, drain, sync, coll, repeats=0, target_seconds=1.0):
    """The timed region repeated (same count on every rank); returns the list of region times (seconds, MAX over ranks)."""
    regions = [timed_region(steps, step, drain, sync, coll)]
    n = repeat_count(regions[0], repeats, target_seconds)
    while len(regions) < n:
        regions.append(timed_region(steps, step, drain, sync, coll))
    return regions


class StepLoop:
    """K slots (contexts / streams) take the steps in turn, alternating between two batches; a slot's previous collective is
    waited for (stream-level) before the slot is reused.  `run(slot, which_batch)` does the slot's work and returns the
    collective's handle (an object with .wait()) or None."""

    def __init__(self, n_slots, run, enter=None):
        self.k, self.run, self.enter = max(1, n_slots), run, enter
        self.pending = [None] * self.k
        self.count = 0

    def step(self):
        j = self.count % self.k
        which = (self.count // self.k) % 2
        self.count += 1
        ctx = self.enter(j) if self.enter else None
        if ctx is not None:
            ctx.__enter__()
        try:
            if self.pending[j] is not None:
                self.pending[j].wait()  # rec[j] / gathered[j] are free again
                self.pending[j] = None
            self.pending[j] = self.run(j, which)
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)

    def drain(self):
        for j in range(self.k):
            if self.pending[j] is not None:
                ctx = self.enter(j) if self.enter else None
                if ctx is not None:
                    ctx.__enter__()
                try:
                    self.pending[j].wait()
                finally:
                    if ctx is not None:
                        ctx.__exit__(None, None, None)
                self.pending[j] = None

    @property
    def last_slot(self):
        return (self.count - 1) % self.k
