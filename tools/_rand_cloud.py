import sys, numpy as np
sys.path.insert(0, '.')
from feature_extraction_amd import capi
capi.load()
rng = np.random.default_rng(5)
for n, lim in ((28800, {}), (28800, dict(max_ring_points=8192)), (6000, {}), (28800, dict(max_ring_points=28800, max_candidates=3500))):
    pts = np.zeros((n, 4), np.float32)
    pts[:, 0] = rng.uniform(0, 100, n); pts[:, 1] = rng.uniform(-50, 50, n); pts[:, 2] = rng.uniform(-1.5, 4, n)
    for preset in ("launch", "default"):
        ctx = capi.Context(capi.params(preset), capi.limits(1, 28800, **lim))
        got = ctx.process_host([pts])[0]
        print(n, lim, preset, "flags", hex(got["flags"]), "K", got["n_keypoints"], "nf", len(got["filtered"]), flush=True)
        ctx.close()
