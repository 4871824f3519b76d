import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def fxlib():
    from feature_extraction_amd import build, capi
    build.build()
    return capi.load()


@pytest.fixture(scope="session")
def fxtestlib(fxlib):
    """lib/libfx_hip_test.so (-DFX_TEST_HOOKS): the only build that exports the header's fx_test_* section."""
    from feature_extraction_amd import build, capi
    build.build_test_hooks()
    return capi.load_test()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py
    oracle_py.load()
    return oracle_py


@pytest.fixture
def fx_hooks(fxlib, monkeypatch):
    """The test build of the library (lib/libfx_hip_test.so, -DFX_TEST_HOOKS) for the duration of a test: contexts created
    inside read the environment hooks the returned function sets (FX_FRONT=0: the separate front kernels, FX_FRONT_FORCE=1 / 2:
    every scan through k_front_redo / k_slow, FX_MERGE_BIG_CAP, FX_DENSE_LDS_KEYS, FX_DENSE_WON_POINTS, FX_TIER_MIN_GRID, ...).  The product
    library has none of them."""
    from feature_extraction_amd import capi
    with capi.test_hooks():
        yield lambda **env: [monkeypatch.setenv(k, str(v)) for k, v in env.items()]
