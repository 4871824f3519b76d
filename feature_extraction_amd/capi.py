"""ctypes binding of include/fx.h (libfx_hip.so).

Plumbing for tests/ and bench.py only: the product is the C-ABI library and the C++ host
classes above it (csrc/fx_node.hpp).  Loading fails loudly when the library is missing —
there is no Python/CPU fallback for the hot path.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfx_hip.so")

FX_DESC_BINS = 1980
FX_DESC_FLOATS = 1989
FX_FEATURE_RECORD_BYTES = 7984
FX_OK = 0
FX_ERR_NO_DEVICE = 2
FX_IN_DEVICE, FX_OUT_HOST, FX_OUT_DEBUG, FX_OUT_CLOUDS = 1, 2, 4, 8
FX_FLAG_RING_OVERFLOW, FX_FLAG_CAND_OVERFLOW, FX_FLAG_KP_OVERFLOW, FX_FLAG_NBR_OVERFLOW = 0x1, 0x2, 0x4, 0x8
FX_FLAG_NAMES = {0x1: "RING_OVERFLOW", 0x2: "CAND_OVERFLOW", 0x4: "KP_OVERFLOW", 0x8: "NBR_OVERFLOW",
                 0x10: "TOTAL_KP_OVERFLOW", 0x20: "KPC_OVERFLOW"}
FX_N_STAGES = 9
STAGE_NAMES = ("k_prep", "k_bucket", "k_rings_runs", "k_rings_large", "k_merge", "k_gather", "k_desc_group", "k_desc_mid",
               "k_desc_rare")
# stages that are one kernel launch (eligible as the roofline line's dominant kernel: their HIP-event span is that kernel)
SINGLE_LAUNCH_STAGES = ("k_prep", "k_bucket", "k_rings_runs", "k_gather", "k_desc_group", "k_desc_mid")
# kernels launched inside each timed stage (rocprofv3 / PMC rows are per kernel name)
STAGE_KERNELS = {"k_prep": ("k_prep", "k_front", "k_front_ab"),  # (k_front: stages 0-4 of scans that fit its LDS tables, in one launch)
                 "k_bucket": ("k_bucket", "k_bucket_many"),  # (k_bucket_many: sensors of more than 24 rings)
                 "k_rings_large": ("k_rings_runs2", "k_rings_large"),  # (k_rings_runs2: sensors of more than 16 rings)
                 "k_merge": ("k_merge_small", "k_merge_big", "k_merge_huge", "k_front_redo", "k_slow"),
                 "k_desc_mid": ("k_desc_mid",),
                 "k_gather": ("k_gather", "k_rng_ord"),  # (k_rng_ord only when several workgroups share a scan: small batches)
                 "k_desc_rare": ("k_dense_sort", "k_dense_density", "k_dense_finish")}


class FxParams(C.Structure):
    _fields_ = [("cloud_leveling", C.c_int32),
                ("x_min", C.c_double), ("x_max", C.c_double),
                ("y_min", C.c_double), ("y_max", C.c_double),
                ("z_min", C.c_double), ("z_max", C.c_double),
                ("cluster_tolerance", C.c_double),
                ("cluster_min_count", C.c_int32), ("cluster_max_count", C.c_int32),
                ("cluster_radius_threshold", C.c_double),
                ("number_detection_channels", C.c_int32),
                ("estimate_descriptors", C.c_int32),
                ("descriptor_radius", C.c_double),
                ("n_rings", C.c_int32), ("el0_deg", C.c_double), ("el_step_deg", C.c_double),
                ("secondary_max", C.c_int32)]


class FxLimits(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in
                ("max_batch", "max_points", "max_ring_points", "max_ring_candidates", "max_candidates",
                 "max_keypoints", "max_neighbors", "max_total_keypoints", "max_kpc_points", "max_dense_points", "max_overflow_points")]


class FxScanDesc(C.Structure):
    _fields_ = [("points", C.c_void_p), ("n_points", C.c_uint32), ("stride_bytes", C.c_uint32),
                ("roll", C.c_double), ("pitch", C.c_double)]


_U32P, _F32P, _I32P = C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.POINTER(C.c_int32)


class FxBatchView(C.Structure):
    _fields_ = [("batch", C.c_uint32), ("max_points", C.c_uint32), ("max_keypoints", C.c_uint32),
                ("max_candidates", C.c_uint32), ("max_kpc_points", C.c_uint32), ("total_keypoints", C.c_uint32),
                ("d_n_keypoints", C.c_void_p), ("d_kp_offset", C.c_void_p), ("d_keypoints", C.c_void_p),
                ("d_descriptors", C.c_void_p), ("d_flags", C.c_void_p), ("d_n_filtered", C.c_void_p),
                ("d_filtered", C.c_void_p), ("d_n_kpc", C.c_void_p), ("d_kpc", C.c_void_p),
                ("h_n_keypoints", _U32P), ("h_kp_offset", _U32P), ("h_keypoints", _F32P),
                ("h_descriptors", _F32P), ("h_flags", _U32P), ("h_n_filtered", _U32P), ("h_filtered", _F32P),
                ("h_n_kpc", _U32P), ("h_kpc", _F32P),
                ("h_n_candidates", _U32P), ("h_candidates", _F32P), ("h_cand_size", _U32P),
                ("h_cand_keypoint", _I32P), ("h_kpc_cand", _U32P), ("h_kp_size", _U32P),
                ("h_kp_neighbors", _U32P)]


class FxPc2Layout(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in
                ("point_step", "offset_x", "offset_y", "offset_z", "offset_intensity", "is_bigendian")]


class FxTimings(C.Structure):
    _fields_ = [("ms", C.c_float * FX_N_STAGES), ("total_ms", C.c_float), ("k_prep_exec_ms", C.c_float)]


class FxStageBytes(C.Structure):
    _fields_ = [("read", C.c_double * FX_N_STAGES), ("written", C.c_double * FX_N_STAGES)]


class FxSynthCfg(C.Structure):
    _fields_ = [("n_rings", C.c_uint32), ("n_az", C.c_uint32), ("el0_deg", C.c_double), ("el_step_deg", C.c_double),
                ("n_poles", C.c_uint32), ("pole_radius", C.c_double), ("pole_height", C.c_double),
                ("x_lo", C.c_double), ("x_hi", C.c_double), ("y_lo", C.c_double), ("y_hi", C.c_double),
                ("sensor_height", C.c_double), ("wall_radius", C.c_double), ("seed", C.c_uint64)]


# every symbol include/fx.h declares (tests/test_capi_symbols.py checks the list against the header)
FX_HEADER_VERSION = (0 << 16) | 7  # the include/fx.h these ctypes structures mirror
EXPORTS = ("fx_version", "fx_check_abi", "fx_status_str", "fx_last_error", "fx_params_default", "fx_params_launch",
           "fx_limits_default", "fx_limits_sparse", "fx_create", "fx_destroy", "fx_set_stream", "fx_get_stream", "fx_set_graph_batch", "fx_set_batches_in_flight", "fx_set_profiling", "fx_set_profiling_stages", "fx_get_timings",
           "fx_get_stage_bytes", "fx_get_limits", "fx_process_batch", "fx_synchronize", "fx_pack_features", "fx_pack_keypoint_records", "fx_keypoint_block_bytes", "fx_pack_keypoint_block",
           "fx_rotation_from_roll_pitch", "fx_sc3d_tables", "fx_sc3d_xaxis", "fx_synth_cfg_vlp16",
           "fx_synth_scan", "fx_unpack_pointcloud2", "fx_pack_pointxyzi")
# the header's FX_TEST_HOOKS section: exported by lib/libfx_hip_test.so only
TEST_EXPORTS = ("fx_test_sort_replay", "fx_test_sort_replay_ranked", "fx_test_sort_replay_lists", "fx_test_sort_replay_device",
                "fx_test_elevation_device", "fx_test_within_device")

_lib = None
_libs = {}
# lib/libfx_hip_test.so: the same sources compiled with -DFX_TEST_HOOKS (environment hooks that push work through the rarely
# used tiers and kernels: FX_FRONT, FX_FRONT_FORCE, FX_MERGE_BIG_CAP, ...).  The product library has none of them.
TEST_LIB_PATH = os.path.join(_HERE, "lib", "libfx_hip_test.so")


class test_hooks:
    """Context manager: inside it load() (and so Context, params, ...) uses the test build of the library."""

    def __enter__(self):
        global _lib, LIB_PATH
        self._saved = (_lib, LIB_PATH)
        LIB_PATH = TEST_LIB_PATH
        _lib = _libs.get(LIB_PATH)
        try:
            return load()
        except BaseException:  # (a missing / mismatched test library must not leave the process pointing at it)
            _lib, LIB_PATH = self._saved
            raise

    def __exit__(self, *exc):
        global _lib, LIB_PATH
        _lib, LIB_PATH = self._saved
        return False


def load_test():
    """The test build of the library (FX_TEST_HOOKS: the fx_test_* entry points and the environment hooks) without making it
    the process's default."""
    with test_hooks() as lib:
        return lib


def load():
    """Load libfx_hip.so; raises (never falls back) if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if LIB_PATH in _libs:
        _lib = _libs[LIB_PATH]
        return _lib
    try:
        # When torch is going to be used in the same process (tests, bench.py) it must be imported
        # before libfx_hip.so is loaded: both link libamdhip64 and the process must end up with ONE HIP
        # runtime; with the other order the second runtime to initialise finds no device.
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). The hot path has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.fx_version.restype = C.c_uint32
    # the ABI guard before anything else: these structures must be the library's (include/fx.h fx_check_abi)
    lib.fx_last_error.restype = C.c_char_p
    lib.fx_check_abi.argtypes = [C.c_uint32, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]
    if lib.fx_check_abi(FX_HEADER_VERSION, C.sizeof(FxParams), C.sizeof(FxLimits), C.sizeof(FxScanDesc), C.sizeof(FxBatchView)) != FX_OK:
        raise RuntimeError(f"{LIB_PATH}: {lib.fx_last_error().decode()}")
    lib.fx_status_str.restype = C.c_char_p
    lib.fx_status_str.argtypes = [C.c_int]
    lib.fx_last_error.restype = C.c_char_p
    lib.fx_params_default.argtypes = [C.POINTER(FxParams)]
    lib.fx_params_launch.argtypes = [C.POINTER(FxParams)]
    lib.fx_limits_default.argtypes = [C.POINTER(FxLimits), C.c_uint32, C.c_uint32]
    lib.fx_limits_sparse.argtypes = [C.POINTER(FxLimits), C.c_uint32, C.c_uint32]
    lib.fx_create.argtypes = [C.POINTER(FxParams), C.POINTER(FxLimits), C.c_int, C.POINTER(C.c_void_p)]
    lib.fx_create.restype = C.c_int
    lib.fx_destroy.argtypes = [C.c_void_p]
    lib.fx_destroy.restype = None
    lib.fx_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.fx_set_profiling.argtypes = [C.c_void_p, C.c_int]
    lib.fx_set_profiling_stages.argtypes = [C.c_void_p, C.c_uint32]
    lib.fx_set_graph_batch.argtypes = [C.c_void_p, C.c_uint32]
    lib.fx_set_batches_in_flight.argtypes = [C.c_void_p, C.c_uint32]
    lib.fx_get_stream.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    lib.fx_get_stream.restype = C.c_int
    lib.fx_get_timings.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(FxTimings)]
    lib.fx_get_limits.argtypes = [C.c_void_p, C.POINTER(FxLimits)]
    lib.fx_get_stage_bytes.argtypes = [C.c_void_p, C.POINTER(FxStageBytes)]
    lib.fx_process_batch.argtypes = [C.c_void_p, C.POINTER(FxScanDesc), C.c_uint32, C.c_uint32,
                                     C.POINTER(FxBatchView)]
    lib.fx_process_batch.restype = C.c_int
    lib.fx_synchronize.argtypes = [C.c_void_p]
    lib.fx_pack_features.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    lib.fx_pack_keypoint_records.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    lib.fx_keypoint_block_bytes.argtypes = [C.c_uint32, C.c_uint32]
    lib.fx_keypoint_block_bytes.restype = C.c_size_t
    lib.fx_pack_keypoint_block.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
    lib.fx_rotation_from_roll_pitch.argtypes = [C.c_double, C.c_double, _F32P]
    lib.fx_rotation_from_roll_pitch.restype = None
    lib.fx_sc3d_tables.argtypes = [C.c_double, _F32P, _F32P, _F32P, _F32P]
    lib.fx_sc3d_tables.restype = None
    lib.fx_sc3d_xaxis.argtypes = [C.c_uint32, _F32P]
    lib.fx_sc3d_xaxis.restype = None
    lib.fx_synth_cfg_vlp16.argtypes = [C.POINTER(FxSynthCfg), C.c_uint64]
    lib.fx_synth_cfg_vlp16.restype = None
    lib.fx_synth_scan.argtypes = [C.POINTER(FxSynthCfg), _F32P, C.c_uint32]
    lib.fx_synth_scan.restype = C.c_uint32
    lib.fx_unpack_pointcloud2.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(FxPc2Layout), C.c_void_p]
    lib.fx_pack_pointxyzi.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, _U32P]
    if hasattr(lib, "fx_test_sort_replay"):  # the test build (-DFX_TEST_HOOKS)
        lib.fx_test_sort_replay.argtypes = [_U32P, C.c_uint32, _U32P]
        lib.fx_test_sort_replay.restype = None
        lib.fx_test_sort_replay_ranked.argtypes = [_U32P, C.c_uint32, _U32P]
        lib.fx_test_sort_replay_ranked.restype = None
        lib.fx_test_sort_replay_lists.argtypes = [_U32P, C.c_uint32, _U32P]
        lib.fx_test_sort_replay_lists.restype = None
        lib.fx_test_sort_replay_device.argtypes = [C.c_int, _U32P, C.c_uint32, C.c_uint32, _U32P]
        lib.fx_test_elevation_device.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.fx_test_within_device.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p]
    _lib = lib
    _libs[LIB_PATH] = lib
    return lib


class FxError(RuntimeError):
    pass


def check(status):
    if status != FX_OK:
        lib = load()
        raise FxError(f"fx status {status} ({lib.fx_status_str(status).decode()}): {lib.fx_last_error().decode()}")


def params(preset="default", **overrides):
    """fx_params for a named preset: 'default' (ref: node.cpp:9-34) or 'launch'
    (ref: launch/keypoint_playback.launch:17-33), with keyword overrides."""
    lib = load()
    p = FxParams()
    {"default": lib.fx_params_default, "launch": lib.fx_params_launch}[preset](C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def limits(max_batch, max_points, sparse=False, **overrides):
    """fx_limits_default (or, sparse=True, fx_limits_sparse: small dense-tier pools, for VLP-16-class workloads) with overrides."""
    lib = load()
    l = FxLimits()
    (lib.fx_limits_sparse if sparse else lib.fx_limits_default)(C.byref(l), max_batch, max_points)
    for k, v in overrides.items():
        if not hasattr(l, k):
            raise AttributeError(k)
        setattr(l, k, v)
    return l


def synth_cfg(seed, **overrides):
    lib = load()
    c = FxSynthCfg()
    lib.fx_synth_cfg_vlp16(C.byref(c), seed)
    for k, v in overrides.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


def synth_scan(cfg):
    """One synthetic scan as an [N,4] float32 array (x, y, z, 0), firing order."""
    lib = load()
    n = cfg.n_rings * cfg.n_az
    out = np.zeros((n, 4), np.float32)
    got = lib.fx_synth_scan(C.byref(cfg), out.ctypes.data_as(_F32P), n)
    assert got == n
    return out


def _np(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0 or not ptr:
        return np.zeros(shape, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).reshape(shape).copy()


class Context:
    """Thin owner of an fx_ctx."""

    def __init__(self, p, lim, device=0):
        self.lib = load()
        self.params, self.limits = p, lim
        self.handle = C.c_void_p()
        check(self.lib.fx_create(C.byref(p), C.byref(lim), device, C.byref(self.handle)))
        got = FxLimits()
        check(self.lib.fx_get_limits(self.handle, C.byref(got)))
        self.limits = got

    def close(self):
        if self.handle:
            self.lib.fx_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        check(self.lib.fx_set_stream(self.handle, C.c_void_p(stream_ptr)))

    def stream_ptr(self):
        """The hipStream_t the context launches on, as an integer (torch.cuda.ExternalStream(ptr) wraps it)."""
        p = C.c_void_p()
        check(self.lib.fx_get_stream(self.handle, C.byref(p)))
        return p.value or 0

    def set_batches_in_flight(self, n):
        """Launch-policy hint: how many contexts the caller keeps busy on this device at a time (results never depend on it)."""
        check(self.lib.fx_set_batches_in_flight(self.handle, int(n)))

    def set_graph_batch(self, max_batch):
        check(self.lib.fx_set_graph_batch(self.handle, int(max_batch)))

    def set_profiling(self, depth, stages=None):
        """depth > 0: keep HIP-event timings of the last `depth` batches; stages: names to time (default all)."""
        check(self.lib.fx_set_profiling(self.handle, int(depth)))
        mask = 0xffffffff if stages is None else sum(1 << STAGE_NAMES.index(n) for n in stages)
        check(self.lib.fx_set_profiling_stages(self.handle, mask))

    def timings(self, back=0):
        """Per-kernel device ms (HIP events on the launch stream) of the batch `back` calls ago."""
        t = FxTimings()
        check(self.lib.fx_get_timings(self.handle, back, C.byref(t)))
        self.last_k_prep_exec_ms = t.k_prep_exec_ms
        return {STAGE_NAMES[i]: t.ms[i] for i in range(FX_N_STAGES)}, t.total_ms

    def front_active(self):
        """True when the last batch went through the fused front kernel (k_front: stages 0-4 in one launch)."""
        self.lib.fx_debug_front.argtypes = [C.c_void_p]
        self.lib.fx_debug_front.restype = C.c_int
        return bool(self.lib.fx_debug_front(self.handle))

    def stage_bytes(self):
        """Algorithmic bytes (read, written) per stage of the last batch: {stage: (read, written)}."""
        sb = FxStageBytes()
        check(self.lib.fx_get_stage_bytes(self.handle, C.byref(sb)))
        return {STAGE_NAMES[i]: (sb.read[i], sb.written[i]) for i in range(FX_N_STAGES)}

    def pack_keypoint_records(self, dst_device_ptr, rec_keypoints):
        check(self.lib.fx_pack_keypoint_records(self.handle, C.c_void_p(dst_device_ptr), rec_keypoints))

    def pack_keypoint_block(self, dst_device_ptr, max_scans, max_total_keypoints):
        """The last batch's keypoints as one compact block (fx_pack_keypoint_block; layout: sharding.unpack_block)."""
        check(self.lib.fx_pack_keypoint_block(self.handle, C.c_void_p(dst_device_ptr), int(max_scans), int(max_total_keypoints)))

    def synchronize(self):
        check(self.lib.fx_synchronize(self.handle))

    def make_descs(self, ptrs, counts, stride_bytes=16, roll=0.0, pitch=0.0):
        n = len(ptrs)
        arr = (FxScanDesc * n)()
        for i in range(n):
            arr[i].points = ptrs[i]
            arr[i].n_points = int(counts[i])
            arr[i].stride_bytes = stride_bytes
            arr[i].roll = roll[i] if np.ndim(roll) else roll
            arr[i].pitch = pitch[i] if np.ndim(pitch) else pitch
        return arr

    def process_raw(self, descs, batch, flags):
        view = FxBatchView()
        check(self.lib.fx_process_batch(self.handle, descs, batch, flags, C.byref(view)))
        return view

    def process_host(self, scans, roll=0.0, pitch=0.0, debug=True):
        """scans: list of [N,4] (or [N,8]) float32 arrays on the host.  Returns a list of dicts."""
        scans = [np.ascontiguousarray(s, dtype=np.float32) for s in scans]
        stride = scans[0].shape[1] * 4 if scans else 16
        descs = self.make_descs([s.ctypes.data for s in scans], [s.shape[0] for s in scans], stride, roll, pitch)
        flags = FX_OUT_HOST | FX_OUT_CLOUDS | (FX_OUT_DEBUG if debug else 0)
        v = self.process_raw(descs, len(scans), flags)
        return self.unpack(v, debug)

    def unpack(self, v, debug=True):
        B = v.batch
        L = self.limits
        n_kp = _np(v.h_n_keypoints, (B,), np.uint32)
        off = _np(v.h_kp_offset, (B + 1,), np.uint32)
        flags = _np(v.h_flags, (B,), np.uint32)
        n_f = _np(v.h_n_filtered, (B,), np.uint32)
        n_kpc = _np(v.h_n_kpc, (B,), np.uint32)
        kp = _np(v.h_keypoints, (B, L.max_keypoints, 4), np.float32)
        desc = _np(v.h_descriptors, (v.total_keypoints, FX_DESC_FLOATS), np.float32)
        out = []
        if debug:
            n_c = _np(v.h_n_candidates, (B,), np.uint32)
        for b in range(B):
            d = {"flags": int(flags[b]), "n_keypoints": int(n_kp[b]), "keypoints": kp[b, :n_kp[b]],
                 "descriptors": desc[off[b]:off[b] + n_kp[b]] if len(desc) else np.zeros((0, FX_DESC_FLOATS), np.float32)}
            if v.h_filtered:
                base = C.cast(v.h_filtered, C.c_void_p).value + b * L.max_points * 16
                d["filtered"] = _np(C.cast(base, _F32P), (int(n_f[b]), 4), np.float32)
                base = C.cast(v.h_kpc, C.c_void_p).value + b * L.max_kpc_points * 16
                d["kpc"] = _np(C.cast(base, _F32P), (int(n_kpc[b]), 4), np.float32)
            if debug:
                def row(ptr, width, dtype, n, comps=1):
                    base = C.cast(ptr, C.c_void_p).value + b * width * 4 * comps
                    return _np(C.cast(base, type(ptr)), (n, comps) if comps > 1 else (n,), dtype)
                nc = int(n_c[b])
                d["candidates"] = row(v.h_candidates, L.max_candidates, np.float32, nc, 4)
                d["cand_size"] = row(v.h_cand_size, L.max_candidates, np.uint32, nc)
                d["cand_keypoint"] = row(v.h_cand_keypoint, L.max_candidates, np.int32, nc)
                d["kpc_cand"] = row(v.h_kpc_cand, L.max_kpc_points, np.uint32, int(n_kpc[b]))
                d["kp_size"] = row(v.h_kp_size, L.max_keypoints, np.uint32, int(n_kp[b]))
                d["kp_neighbors"] = row(v.h_kp_neighbors, L.max_keypoints, np.uint32, int(n_kp[b]))
            out.append(d)
        return out
