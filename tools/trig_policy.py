"""How much does the phi / theta policy matter?  PCL evaluates phi = atan2f(...), theta = acosf(...) in fp32 with the local
libm; the oracle and the product evaluate them in fp64 and round once (A.8-14), because two
fp32 libms disagree in the last bit and a neighbour next to a bin edge then lands in another bin — a whole weight of
1.5 .. 31 moves.  This tool measures how many descriptor values that is, on both sides, workload by workload:
  * oracle, glibc's atan2f / acosf (oracle_py.TRIG_LIBM_F32) against the oracle's policy — runs anywhere;
  * product, the device's atan2f / acosf with no exact re-evaluation (lib/libfx_hip_trigf32.so, -DFX_TRIG_LITERAL_F32:
    feature_extraction_amd/build.py build_trig_literal) against the product's policy — needs the GPU.
Workloads: bench scans (VLP-16, 64 poles), VLP-16 scenes of 256 poles, and the five golden fixtures (incl. one 64 x 2048 and
one 128 x 2048 scan, whose support sets are 10-100 times larger: bin-edge hits proportionally likelier).
usage: python tools/trig_policy.py [n_bench_scans=32] [n_pole_scans=8]      (prints a table; tee it into profiles/)"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from feature_extraction_amd import build, capi  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
from tests import util  # noqa: E402

n_bench = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n_poles = int(sys.argv[2]) if len(sys.argv) > 2 else 8


def moved(a, b):
    """a, b: [K, 1989] descriptors under the two policies.  Values that differ by more than 1e-5, non-empty values, descriptors
    touched, descriptors; and the (row, bin) pairs that moved."""
    a, b = np.nan_to_num(a), np.nan_to_num(b)
    d = np.abs(a - b) > 1e-5
    nz = (a != 0) | (b != 0)
    return int(d.sum()), int(nz.sum()), int(d.any(axis=1).sum()), len(a), np.argwhere(d)


def workloads():
    p = capi.params("launch")
    yield "bench scans (VLP-16, 64 poles)", p, capi.limits(n_bench, 28800), [util.vlp16_scan(1000 + b) for b in range(n_bench)], 0.02, -0.015
    yield ("VLP-16, 256 poles", p, capi.limits(n_poles, 28800, max_candidates=3500, max_keypoints=1024, max_total_keypoints=n_poles * 1024, max_kpc_points=8192),
           [util.vlp16_scan(7 + b, n_poles=256) for b in range(n_poles)], 0.02, -0.015)
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz"))):
        name = os.path.basename(path)[:-4]
        pp, lim, pts, roll, pitch = util.golden_case(np.load(path), name)
        yield f"fixture {name}", pp, lim, [pts], roll, pitch


try:
    import torch
    have_gpu = torch.cuda.is_available()
except Exception:
    have_gpu = False
libs = None
if have_gpu:
    libs = ("libfx_hip.so", os.path.basename(build.build_trig_literal()))
print(f"{'workload':52s} {'side':8s} {'moved':>7s} {'non-empty values':>17s} {'descriptors touched':>20s}")
for name, p, lim, scans, roll, pitch in workloads():
    res = [np.concatenate([O.run(p, s, roll=roll, pitch=pitch, trig=t)["descriptors"] for s in scans]) for t in (O.TRIG_F64_ROUNDED, O.TRIG_LIBM_F32)]
    d, nz, rows, total, where = moved(*res)
    print(f"{name:52s} {'oracle':8s} {d:7d} {nz:17d} {rows:9d} of {total:6d}   (glibc atan2f / acosf vs fp64 rounded once)")
    for r, b in where[:8]:
        print(f"    descriptor {r} bin {b} (azimuth {b // 165} elevation {b % 165 // 15} radius {b % 15}): {res[0][r, b]!r} -> {res[1][r, b]!r}")
    if libs:
        out = []
        for lib in libs:
            saved = capi.LIB_PATH, capi._lib
            capi.LIB_PATH, capi._lib = os.path.join(os.path.dirname(saved[0]), lib), None
            try:
                ctx = capi.Context(p, lim)
                got = ctx.process_host(scans, roll=roll, pitch=pitch)
                assert all(g["flags"] == 0 for g in got)
                out.append(np.concatenate([g["descriptors"] for g in got]))
                ctx.close()
            finally:
                capi.LIB_PATH, capi._lib = saved
        d, nz, rows, total, where = moved(*out)
        print(f"{name:52s} {'product':8s} {d:7d} {nz:17d} {rows:9d} of {total:6d}   (device atan2f / acosf vs exact next to a bin edge)")
        for r, b in where[:8]:
            print(f"    descriptor {r} bin {b} (azimuth {b // 165} elevation {b % 165 // 15} radius {b % 15}): {out[0][r, b]!r} -> {out[1][r, b]!r}")
if not have_gpu:
    print("product side skipped: no GPU here")
