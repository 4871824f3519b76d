"""Builds lib/libfx_hip.so with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libfx_hip.so")
# the same sources with -DFX_TEST_HOOKS: the environment hooks tests (and A/B measurements) use to push work through the
# rarely used tiers and kernels — the product library has none of them compiled in
LIB_TEST = os.path.join(HERE, "lib", "libfx_hip_test.so")
SOURCES = ["fx_kernels.hip", "fx_api.cpp", "fx_host.cpp"]
HEADERS = ["fx_device.h", "fx_sort_replay.h", os.path.join("..", "..", "include", "fx.h")]
# -ffp-contract=off: the numerics contract forbids FMA contraction (results must follow
# PCL/FLANN/Eigen operation order); fp32 divide/sqrt stay at hipcc's correctly rounded default.
# -O2, not -O3: the kernels are big (k_front 60 KB, k_rings_runs 42 KB of code against a 64 KB instruction cache that two CUs
# share) and -O3's extra unrolling and inlining costs more in instruction fetches than it saves: measured in one gpurun call,
# alternating, the headline 2.22e6 scans/s against 2.19e6 (profiles/r04_front_experiments.md).
FLAGS = ["--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-Wl,-Bsymbolic"]


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm; the hot path has no non-HIP build)")


def stale(lib=None):
    lib = lib or LIB
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def _link(lib, extra, verbose):
    """Compile to a temporary name and rename into place: a process that polls for the library (the other ranks of a
    multi-GPU bench wait for local rank 0's build) never maps a half-written file."""
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    tmp = f"{lib}.{os.getpid()}.tmp"
    cmd = [hipcc()] + FLAGS + extra + ["-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, lib)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def build_test_hooks(force=False, verbose=False):
    if force or stale(LIB_TEST):
        _link(LIB_TEST, ["-DFX_TEST_HOOKS"], verbose)
    return LIB_TEST


LIB_TRIG = os.path.join(HERE, "lib", "libfx_hip_trigf32.so")


def build_trig_literal(force=False, verbose=False):
    """Measurement build (-DFX_TRIG_LITERAL_F32): phi / theta as PCL writes them — the device's atan2f / acosf, no exact
    re-evaluation next to a bin edge.  tools/trig_policy.py and tests/test_gpu_trig_policy.py compare it with the product."""
    if force or stale(LIB_TRIG):
        _link(LIB_TRIG, ["-DFX_TRIG_LITERAL_F32"], verbose)
    return LIB_TRIG


LIB_SKIPEPS = os.path.join(HERE, "lib", "libfx_hip_skipeps.so")


def build_skip_epsilon(force=False, verbose=False):
    """Measurement build (-DFX_SKIP_EPSILON): 3DSC skips a neighbour at d^2 < FLT_EPSILON (SURVEY.md A.8-6's reading) instead of
    < numeric_limits<float>::min() — the one reading of PCL the oracle's policy switches show to be live.
    tests/test_gpu_skip_policy.py holds it against the oracle's FXO_POLICY_SKIP_EPSILON."""
    if force or stale(LIB_SKIPEPS):
        _link(LIB_SKIPEPS, ["-DFX_SKIP_EPSILON"], verbose)
    return LIB_SKIPEPS


def build_variant(name, defines, force=False, verbose=False):
    """A measurement build lib/libfx_hip_<name>.so with extra -D flags (tools/bench_lib.py, tools/*_stamps.py): diagnostic, never
    the product."""
    lib = os.path.join(HERE, "lib", f"libfx_hip_{name}.so")
    if force or stale(lib):
        _link(lib, list(defines), verbose)
    return lib


CLI = os.path.join(HERE, "bin", "fx_cli")
CLI_SOURCES = ["fx_cli.cpp", "fx_node.hpp", "fx_pcd.hpp"]


def build_cli(force=False, verbose=False):
    """C++ host front end (csrc/fx_node.hpp mirror of the reference class + .pcd CLI) over the C-ABI."""
    deps = [os.path.join(CSRC, s) for s in CLI_SOURCES] + [LIB]
    if not force and os.path.exists(CLI) and all(os.path.getmtime(d) <= os.path.getmtime(CLI) for d in deps):
        return CLI
    os.makedirs(os.path.dirname(CLI), exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++17", "-o", CLI, os.path.join(CSRC, "fx_cli.cpp"), "-L" + os.path.dirname(LIB), "-lfx_hip",
           "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath," + os.path.dirname(LIB)]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return CLI


MULTI = os.path.join(HERE, "bin", "fx_multi_cli")
SELFTEST = os.path.join(HERE, "bin", "fx_shard_selftest")


def _newer(target, deps):
    return os.path.exists(target) and all(os.path.getmtime(d) <= os.path.getmtime(target) for d in deps)


def build_multi(force=False, verbose=False):
    """C++ multi-GPU driver (csrc/fx_multi.hpp: one thread + context per device, RCCL all-gather of keypoint records)
    and the CPU self-test of its sharding plan / record layout (csrc/fx_shard.hpp)."""
    os.makedirs(os.path.dirname(MULTI), exist_ok=True)
    deps = [os.path.join(CSRC, s) for s in ("fx_shard_selftest.cpp", "fx_shard.hpp")]
    if force or not _newer(SELFTEST, deps):
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-o", SELFTEST, os.path.join(CSRC, "fx_shard_selftest.cpp")]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    deps = [os.path.join(CSRC, s) for s in ("fx_multi_cli.cpp", "fx_multi.hpp", "fx_shard.hpp")] + [LIB]
    if force or not _newer(MULTI, deps):
        rocm_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc()))), "lib")
        cmd = [hipcc(), "-O2", "-std=c++17", "-o", MULTI, os.path.join(CSRC, "fx_multi_cli.cpp"), "-L" + os.path.dirname(LIB),
               "-lfx_hip", "-L" + rocm_lib, "-lrccl", "-pthread", "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath," + os.path.dirname(LIB),
               "-Wl,-rpath," + rocm_lib]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return MULTI


BATCHER = os.path.join(HERE, "bin", "fx_batcher_cli")
BATCHER_TEST = os.path.join(HERE, "bin", "fx_batcher_cli_test")


def build_batcher(force=False, verbose=False, test_hooks=False):
    """Streaming front end (csrc/fx_batcher.hpp: producers push scans, one consumer batches whatever has arrived) and its
    simulated-sensors driver.  test_hooks: the same driver linked against the TEST build of the library (its environment
    hooks: FX_FAIL_AFTER_ENQUEUE makes a batch fail)."""
    exe, lib = (BATCHER_TEST, build_test_hooks()) if test_hooks else (BATCHER, LIB)
    deps = [os.path.join(CSRC, s) for s in ("fx_batcher_cli.cpp", "fx_batcher.hpp", "fx_node.hpp")] + [lib]
    if force or not _newer(exe, deps):
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-pthread", "-o", exe, os.path.join(CSRC, "fx_batcher_cli.cpp"),
               "-L" + os.path.dirname(lib), "-l:" + os.path.basename(lib), "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath," + os.path.dirname(lib)]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return exe


ROS_MOCK_NODE = os.path.join(HERE, "bin", "fx_ros_mock_node")


def build_ros_mock(force=False, verbose=False):
    """The ROS1 shell (ros/feature_extraction_node.cpp) compiled against the stand-in headers of tests/ros_mock — test
    infrastructure, NOT roscpp (ROS is not installable here); the real build goes through ros/CMakeLists.txt."""
    root = os.path.dirname(HERE)
    mock = os.path.join(root, "tests", "ros_mock")
    src = os.path.join(root, "ros", "feature_extraction_node.cpp")
    deps = [src, os.path.join(CSRC, "fx_node.hpp"), LIB] + [os.path.join(d, f) for d, _, fs in os.walk(mock) for f in fs]
    if force or not _newer(ROS_MOCK_NODE, deps):
        os.makedirs(os.path.dirname(ROS_MOCK_NODE), exist_ok=True)
        cmd = ["g++", "-O1", "-std=c++17", "-Wall", "-I", mock, "-o", ROS_MOCK_NODE, src, "-L" + os.path.dirname(LIB), "-lfx_hip",
               "-Wl,-rpath," + os.path.dirname(LIB)]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return ROS_MOCK_NODE


def build(force=False, verbose=False):
    if force or stale():
        _link(LIB, [], verbose)
    build_test_hooks(force, verbose)
    build_trig_literal(force, verbose)
    build_skip_epsilon(force, verbose)
    build_cli(force, verbose)
    build_batcher(force, verbose)
    try:  # the multi-GPU driver needs RCCL's development files: without them the library and everything else still build
        build_multi(force, verbose)
    except (subprocess.CalledProcessError, OSError) as e:
        print(f"[fx build] fx_multi_cli not built ({e}); tests/test_gpu_multi.py and tests/test_cpp_sharding.py build it on demand")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
