"""How much does the phi / theta policy matter?  PCL evaluates phi = atan2f(...), theta = acosf(...) in fp32 with the local
libm; the oracle and the product evaluate them in fp64 and round once (A.8-14), because two
fp32 libms disagree in the last bit and a neighbour next to a bin edge then lands in another bin — a whole weight of
1.5 .. 31 moves.  This tool measures how many descriptor values that is, on both sides:
  * oracle, glibc's atan2f / acosf (FXO_TRIG_LIBM_F32=1) against the oracle's policy — runs anywhere;
  * product, the device's atan2f / acosf with no exact re-evaluation (lib/libfx_hip_trigf32.so, built by this tool with
    -DFX_TRIG_LITERAL_F32) against the product's policy — needs the GPU.
usage: python tools/trig_policy.py [n_scans=8]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from feature_extraction_amd import build, capi  # noqa: E402

n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def moved(a, b):
    a, b = np.nan_to_num(a), np.nan_to_num(b)
    d = np.abs(a - b) > 1e-5
    nz = (a != 0) | (b != 0)
    return int(d.sum()), int(nz.sum()), int(d.any(axis=1).sum()), len(a)


def oracle_side(scans, p):
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from oracle import oracle_py as O; from feature_extraction_amd import capi; "
            "p = capi.params('launch'); "
            "out = [O.run(p, capi.synth_scan(capi.synth_cfg(1000 + b)), roll=0.02, pitch=-0.015)['descriptors'] for b in range(%d)]; "
            "np.save(sys.argv[1], np.concatenate(out))") % (ROOT, len(scans))
    res = []
    for env in ({}, {"FXO_TRIG_LIBM_F32": "1"}):
        path = "/tmp/_trig_%d.npy" % len(res)
        subprocess.check_call([sys.executable, "-c", code, path], env=dict(os.environ, **env))
        res.append(np.load(path))
    return moved(res[0], res[1])


def product_side(scans, p):
    lib = os.path.join(os.path.dirname(capi.LIB_PATH), "libfx_hip_trigf32.so")
    cmd = [build.hipcc()] + build.FLAGS + ["-DFX_TRIG_LITERAL_F32", "-o", lib] + [os.path.join(build.CSRC, s) for s in build.SOURCES]
    subprocess.check_call(cmd)
    code = ("import sys, os, numpy as np; sys.path.insert(0, %r); from feature_extraction_amd import capi; "
            "capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), sys.argv[2]); "
            "p = capi.params('launch'); scans = [capi.synth_scan(capi.synth_cfg(1000 + b)) for b in range(%d)]; "
            "ctx = capi.Context(p, capi.limits(len(scans), 28800)); got = ctx.process_host(scans, roll=0.02, pitch=-0.015); "
            "np.save(sys.argv[1], np.concatenate([g['descriptors'] for g in got]))") % (ROOT, len(scans))
    res = []
    for name in ("libfx_hip.so", "libfx_hip_trigf32.so"):
        path = "/tmp/_trigp_%d.npy" % len(res)
        subprocess.check_call([sys.executable, "-c", code, path, name])
        res.append(np.load(path))
    return moved(res[0], res[1])


scans = list(range(n_scans))
p = None
d, nz, rows, total = oracle_side(scans, p)
print(f"oracle, glibc atan2f/acosf vs fp64-rounded-once: {d} of {nz} non-empty descriptor values differ by more than 1e-5 "
      f"({100.0 * d / max(nz, 1):.3f} %), in {rows} of {total} descriptors")
try:
    import torch
    have_gpu = torch.cuda.is_available()
except Exception:
    have_gpu = False
if have_gpu:
    d, nz, rows, total = product_side(scans, p)
    print(f"product, device atan2f/acosf vs exact-next-to-an-edge: {d} of {nz} non-empty descriptor values differ by more than 1e-5 "
          f"({100.0 * d / max(nz, 1):.3f} %), in {rows} of {total} descriptors")
else:
    print("product side skipped: no GPU here")
