"""Builds lib/libfx_hip.so with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libfx_hip.so")
SOURCES = ["fx_kernels.hip", "fx_api.cpp", "fx_host.cpp"]
HEADERS = ["fx_device.h", "fx_sort_replay.h", os.path.join("..", "..", "include", "fx.h")]
# -ffp-contract=off: the numerics contract forbids FMA contraction (results must follow
# PCL/FLANN/Eigen operation order); fp32 divide/sqrt stay at hipcc's correctly rounded default.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared"]


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm; the hot path has no non-HIP build)")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


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


def build(force=False, verbose=False):
    if force or stale():
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        cmd = [hipcc()] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    build_cli(force, verbose)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
