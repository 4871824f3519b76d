"""The C-ABI library loads on a CPU-only box and exports every symbol include/fx.h declares;
without a GPU the hot path refuses to run instead of falling back."""
import ctypes as C
import os
import re

import pytest

from feature_extraction_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(test_section=False):
    """Function names include/fx.h declares: outside (default) or inside its `#ifdef FX_TEST_HOOKS` section."""
    src = open(os.path.join(ROOT, "include", "fx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    m = re.search(r"#ifdef FX_TEST_HOOKS(.*?)#endif", src, flags=re.S)
    assert m, "the header's test section"
    src = m.group(1) if test_section else src[:m.start()] + src[m.end():]
    names = re.findall(r"\b(fx_[a-z0-9_]+)\s*\(", src)
    return sorted(set(n for n in names if not n.endswith("_t")))


def test_header_symbols_exported(fxlib):
    declared = _declared()
    assert len(declared) >= 20
    missing = [n for n in declared if not hasattr(fxlib, n)]
    assert not missing, missing
    assert sorted(capi.EXPORTS) == declared, (set(capi.EXPORTS) ^ set(declared))


def test_test_entry_points_only_in_the_test_build(fxlib, fxtestlib):
    """the fx_test_* entry points (and the k_test_* kernels behind them) are compiled only with -DFX_TEST_HOOKS:
    the header declares them in a guarded section, lib/libfx_hip_test.so exports them, the product library does not."""
    import subprocess
    from feature_extraction_amd import build
    hooks = _declared(test_section=True)
    assert sorted(capi.TEST_EXPORTS) == hooks and len(hooks) >= 6
    assert not [n for n in hooks + list(capi.EXPORTS) if not hasattr(fxtestlib, n)]
    dyn = subprocess.check_output(["nm", "-D", "--defined-only", build.LIB], text=True)
    assert "fx_create" in dyn and "fx_test_" not in dyn, [l for l in dyn.splitlines() if "fx_test_" in l]
    product = open(build.LIB, "rb").read()
    assert b"k_test_" not in product  # (kernel names are in the code object's symbol table)
    assert b"k_test_within" in open(build.build_test_hooks(), "rb").read()


def test_struct_sizes_match_header(fxlib):
    # fx_params: int32 + 6 doubles + double + 2 int32 + double + int32 + int32 + double + int32 + 2 doubles + int32
    assert C.sizeof(capi.FxParams) == 128
    assert C.sizeof(capi.FxLimits) == 11 * 4
    assert C.sizeof(capi.FxScanDesc) == 32
    assert C.sizeof(capi.FxTimings) == (capi.FX_N_STAGES + 2) * 4


def test_version_and_status_strings(fxlib):
    assert fxlib.fx_version() == (0 << 16) | 7 == capi.FX_HEADER_VERSION  # FX_VERSION_MAJOR << 16 | FX_VERSION_MINOR (include/fx.h)
    for code in range(6):
        assert fxlib.fx_status_str(code)
    assert b"no CPU fallback" in fxlib.fx_status_str(capi.FX_ERR_NO_DEVICE)


def test_abi_guard_refuses_other_headers(fxlib):
    """fx_limits grew in 0.4 and the library reads every member — a caller compiled against another header (another
    version, or a struct of another size) is refused by fx_check_abi before it gets to fx_create."""
    sizes = [C.sizeof(capi.FxParams), C.sizeof(capi.FxLimits), C.sizeof(capi.FxScanDesc), C.sizeof(capi.FxBatchView)]
    assert fxlib.fx_check_abi(capi.FX_HEADER_VERSION, *sizes) == capi.FX_OK
    assert fxlib.fx_check_abi((0 << 16) | 3, *sizes) == 1  # FX_ERR_INVALID_ARG
    assert b"0.3" in fxlib.fx_last_error()
    assert fxlib.fx_check_abi(capi.FX_HEADER_VERSION, sizes[0], sizes[1] - 4, sizes[2], sizes[3]) == 1  # 0.5's ten-word fx_limits
    assert b"fx_limits 40" in fxlib.fx_last_error()


def test_product_library_has_no_environment_hooks(fxlib):
    """the shipped library must not change tiers, grids or kernels because of the caller's environment — the hooks
    tests use are compiled only into lib/libfx_hip_test.so (-DFX_TEST_HOOKS)."""
    from feature_extraction_amd import build
    product = open(build.LIB, "rb").read()
    test = open(build.build_test_hooks(), "rb").read()
    for name in (b"FX_MERGE_BIG_CAP", b"FX_FRONT_FORCE", b"FX_FRONT_SPLIT", b"FX_FRONT_STREAM", b"FX_MERGE_SLICES", b"FX_GATHER_COUNTED", b"FX_TIER_MIN_GRID", b"FX_DENSE_LDS_KEYS", b"FX_DENSE_WON_POINTS", b"FX_GRAPH_MAX_BATCH", b"FX_DEBUG_SYNC",
                 b"FX_FAIL_AFTER_ENQUEUE"):
        assert name not in product, name
        assert name in test, name
    assert b"getenv" not in product or b"FX_" not in product[product.find(b"getenv") - 64:product.find(b"getenv") + 64]


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
