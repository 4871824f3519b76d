"""Diagnostic: the differential fuzz of tests/test_gpu_fuzz.py over an arbitrary seed range, both merge tiers.\n  python tools/fuzz_more.py FIRST LAST"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from feature_extraction_amd import capi
from oracle import oracle_py as O
from tests import util
from tests.test_gpu_fuzz import _case
t0 = time.time(); bad = 0; total_k = 0
lo, hi = int(sys.argv[1]), int(sys.argv[2])
for seed in range(lo, hi):
    s, p, roll, pitch, what = _case(seed)
    for tier in ("lds", "large"):
        if tier == "large":
            os.environ["FX_MERGE_BIG_CAP"] = "16"
        else:
            os.environ.pop("FX_MERGE_BIG_CAP", None)
        ctx = capi.Context(p, capi.limits(1, 28800, max_candidates=3500, max_kpc_points=57600, max_keypoints=1024, max_total_keypoints=1024, max_ring_candidates=512))
        got = ctx.process_host([s], roll=roll, pitch=pitch)[0]
        ctx.close()
        if tier == "lds":
            ora = O.run(p, s, roll=roll, pitch=pitch)
        try:
            st = util.compare_scan(got, ora, tag=f"seed {seed} {tier}")
            total_k += st["K"]
        except AssertionError as e:
            bad += 1
            print("MISMATCH", seed, tier, str(e)[:300], what)
print(f"seeds {lo}..{hi}: {bad} mismatches, {total_k} keypoints, {time.time() - t0:.0f} s")
