import sys, os
sys.path.insert(0, "/root/repo")
from feature_extraction_amd import capi
if len(sys.argv) > 1:
    capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), sys.argv[1])
import numpy as np
from oracle import oracle_py as O
from tests import util
import bench
cfg = bench.OTHER_CONFIGS["config3_hdl64_64x2048_batch256"]
s = capi.synth_scan(capi.synth_cfg(10, **cfg["synth"]))
p = capi.params(cfg["preset"], **cfg["params"])
ctx = capi.Context(p, capi.limits(1, len(s), **dict(cfg["limits"], max_total_keypoints=512)))
got = ctx.process_host([s], roll=0.02, pitch=-0.015)[0]
ora = O.run(p, s, roll=0.02, pitch=-0.015)
g, o = got["descriptors"], ora["descriptors"]
bad = np.abs(np.where(np.isnan(o), 0, g) - np.where(np.isnan(o), 0, o)).max(axis=1)
nb = ora["kp_neighbors"]
print("lib", capi.LIB_PATH.split("/")[-1], "rows", len(bad), "bad rows", int((bad > 1e-5).sum()), "their neighbour counts", nb[bad > 1e-5][:20])
