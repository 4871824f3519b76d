"""Diagnostic: differential fuzz on dense many-ring sensors (32 / 64 rings, 512-1024 azimuths): random scenes x random node
parameters against the oracle.  These scans exercise what the VLP-16 fuzz does not: the second run tier, the workgroup ring
tier, the large merge tier, the list and dense descriptor tiers.
  python tools/fuzz_dense.py FIRST LAST [N_AZ [RINGS]]      (e.g. 2048 128: BASELINE config 5's shape, seconds per case)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_extraction_amd import capi
from oracle import oracle_py as O
from tests import util
from tests.test_gpu_fuzz_dense import dense_case

if os.environ.get("FX_USE_TEST_LIB"):  # the test build, so that the FX_* hooks of the environment apply (e.g. FX_DENSE_ONE_FINISH=1)
    capi.test_hooks().__enter__()
else:
    capi.load()
O.load()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
import ctypes as C
bad = total_k = flagged = 0
tiers = np.zeros(16, np.int64)
for seed in range(lo, hi):
    s, p, roll, pitch, lim, what = dense_case(seed, int(sys.argv[3]) if len(sys.argv) > 3 else None, int(sys.argv[4]) if len(sys.argv) > 4 else None)
    R, n_az, over = what["R"], what["n_az"], what["over"]
    ctx = capi.Context(p, lim)
    got = ctx.process_host([s], roll=roll, pitch=pitch)[0]
    cnt = (C.c_uint32 * 16)()
    ctx.lib.fx_debug_counters.argtypes = [C.c_void_p, C.c_void_p]
    capi.check(ctx.lib.fx_debug_counters(ctx.handle, cnt))
    tiers += (np.array(list(cnt)) > 0)
    ctx.close()
    if got["flags"]:
        flagged += 1
        print("flagged", seed, hex(got["flags"]), R, n_az)
        continue
    ora = O.run(p, s, roll=roll, pitch=pitch)
    try:
        st = util.compare_scan(got, ora, tag=f"seed {seed}")
        total_k += st["K"]
    except AssertionError as e:
        bad += 1
        print("MISMATCH", seed, str(e)[:300], dict(R=R, n_az=n_az, over=over))
print(f"dense seeds {lo}..{hi}: {bad} mismatches, {flagged} flagged, {total_k} keypoints, {time.time() - t0:.0f} s")
names = ["second run tier", "big merge", "-", "-", "list rows", "workgroup ring tier", "dense rows", "-", "wave rows", "huge merge",
         "-", "-", "dense rows sorted in global memory"]
print("cases that used: " + ", ".join(f"{n} {int(c)}" for n, c in zip(names, tiers) if n != "-"))
