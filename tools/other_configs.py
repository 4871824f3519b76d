"""Diagnostic: bench.py's other_configs entries alone (configs 3 and 5 with batches in flight).  python tools/other_configs.py [in_flight [library file]]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from feature_extraction_amd import capi
if len(sys.argv) > 2:
    capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), sys.argv[2])
capi.load()
import bench
dev = torch.device("cuda", 0)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 0  # 0: what bench.py uses for the configuration
for name, cfg in bench.OTHER_CONFIGS.items():
    if not name.startswith(("config3", "config5")):
        continue
    r = bench.run_other_config(name, cfg, capi, torch, dev, os.cpu_count() or 1, 0.02, -0.015, in_flight=k or cfg.get("in_flight", 4))
    print(name, f"in flight {k}: {r['scans_per_s']:.0f} scans/s ({r['ms_per_batch']:.3f} ms/batch); one at a time {r['one_at_a_time']['scans_per_s']:.0f}; flags {r['flags_or']}")
