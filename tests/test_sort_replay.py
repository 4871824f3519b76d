"""The kernels' cluster-order replay (csrc/fx_sort_replay.h, host build through the C-ABI test
hooks) against the oracle's literal std::sort(rbegin, rend, bySize) call."""
import ctypes as C

import numpy as np
import pytest

from feature_extraction_amd import capi


def _replay(lib, sizes, ranked):
    s = np.ascontiguousarray(sizes, dtype=np.uint32)
    out = np.zeros(len(s), np.uint32)
    fn = {False: lib.fx_test_sort_replay, True: lib.fx_test_sort_replay_ranked, "lists": lib.fx_test_sort_replay_lists}[ranked]
    fn(s.ctypes.data_as(capi._U32P), len(s), out.ctypes.data_as(capi._U32P))
    return out


@pytest.mark.parametrize("ranked", [False, True, "lists"])
def test_random_sequences(fxtestlib, oracle, ranked):
    rng = np.random.default_rng(11)
    for trial in range(1500):
        n = int(rng.integers(0, 300)) if trial % 12 else int(rng.integers(1000, 4000))
        hi = int(rng.choice([1, 2, 3, 5, 16, 50, 1000]))
        s = rng.integers(1, hi + 1, n)
        assert np.array_equal(_replay(fxtestlib, s, ranked), oracle.sort_by_size_desc(s)), (trial, n, hi)


@pytest.mark.parametrize("ranked", [False, True, "lists"])
def test_structured_sequences(fxtestlib, oracle, ranked):
    for n in (0, 1, 2, 16, 17, 33, 100, 1000, 5000):
        idx = np.arange(n)
        for s in (idx % 60000 + 1, idx[::-1] % 60000 + 1, np.ones(n), np.minimum(idx, idx[::-1]) + 1,
                  (idx * 7919) % 13 + 1):
            assert np.array_equal(_replay(fxtestlib, s, ranked), oracle.sort_by_size_desc(s)), n


@pytest.mark.parametrize("ranked", [False, True, "lists"])
def test_heap_sort_fallback_is_replayed(fxtestlib, oracle, ranked):
    # McIlroy's adversary built against the very std::sort call drives introsort to its depth limit
    for n in (64, 500, 5000, 30000):
        s = oracle.antiqsort(n)
        assert len(np.unique(s)) == n
        assert np.array_equal(_replay(fxtestlib, s, ranked), oracle.sort_by_size_desc(s)), n


@pytest.mark.gpu
def test_wavefront_replay_on_the_device(fxtestlib, oracle):
    """The partition phase as the ring / merge kernels run it (one wavefront, ballots) on random,
    tie-heavy, structured and adversarial sequences of up to 192 clusters."""
    rng = np.random.default_rng(5)
    for n in (17, 18, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 160, 191, 192):
        seqs = []
        for hi in (1, 2, 3, 5, 16, 50, 1000):
            seqs += [rng.integers(1, hi + 1, n) for _ in range(12)]
        idx = np.arange(n)
        seqs += [idx + 1, idx[::-1] + 1, np.ones(n), np.minimum(idx, idx[::-1]) + 1, (idx * 7919) % 13 + 1, oracle.antiqsort(n)]
        s = np.ascontiguousarray(np.stack(seqs), dtype=np.uint32)
        out = np.zeros_like(s)
        capi.check(fxtestlib.fx_test_sort_replay_device(0, s.ctypes.data_as(capi._U32P), len(s), n, out.ctypes.data_as(capi._U32P)))
        for q in range(len(s)):
            assert np.array_equal(out[q], oracle.sort_by_size_desc(s[q])), (n, q)
