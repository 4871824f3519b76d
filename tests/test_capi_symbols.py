"""The C-ABI library loads on a CPU-only box and exports every symbol include/fx.h declares;
without a GPU the hot path refuses to run instead of falling back."""
import ctypes as C
import os
import re

import pytest

from feature_extraction_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "fx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(fx_[a-z0-9_]+)\s*\(", src)
    return sorted(set(n for n in names if not n.endswith("_t")))


def test_header_symbols_exported(fxlib):
    declared = _declared()
    assert len(declared) >= 20
    missing = [n for n in declared if not hasattr(fxlib, n)]
    assert not missing, missing
    assert sorted(capi.EXPORTS) == declared, (set(capi.EXPORTS) ^ set(declared))


def test_struct_sizes_match_header(fxlib):
    # fx_params: int32 + 6 doubles + double + 2 int32 + double + int32 + int32 + double + int32 + 2 doubles + int32
    assert C.sizeof(capi.FxParams) == 128
    assert C.sizeof(capi.FxLimits) == 10 * 4
    assert C.sizeof(capi.FxScanDesc) == 32
    assert C.sizeof(capi.FxTimings) == (capi.FX_N_STAGES + 2) * 4


def test_version_and_status_strings(fxlib):
    assert fxlib.fx_version() == (0 << 16) | 4  # FX_VERSION_MAJOR << 16 | FX_VERSION_MINOR (include/fx.h)
    for code in range(6):
        assert fxlib.fx_status_str(code)
    assert b"no CPU fallback" in fxlib.fx_status_str(capi.FX_ERR_NO_DEVICE)


def test_no_device_means_error_not_fallback(fxlib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.FxError) as e:
        capi.Context(capi.params("default"), capi.limits(4, 1024))
    assert "status 2" in str(e.value)


def test_missing_library_is_loud(monkeypatch, tmp_path):
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", str(tmp_path / "libfx_hip.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        capi.load()
