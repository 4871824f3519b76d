import os, sys
sys.path.insert(0, "/root/repo")
import torch
from feature_extraction_amd import capi
capi.load()
import bench
dev = torch.device("cuda", 0)
mode = sys.argv[1]
name = "config5_dense_128x2048_R2m_batch64"
cfg = bench.OTHER_CONFIGS[name]
keep = []
if mode == "torch_streams":
    keep = [torch.cuda.Stream(device=dev) for _ in range(3)]
if mode == "ctx_churn":
    p = capi.params("launch")
    for _ in range(3):
        cs = [capi.Context(p, capi.limits(64, 28800), device=0) for _ in range(4)]
        for c in cs: c.close()
if mode == "ctx_churn_odd":
    p = capi.params("launch")
    cs = [capi.Context(p, capi.limits(64, 28800), device=0) for _ in range(3)]
    for c in cs: c.close()
if mode == "big_alloc":
    x = torch.empty(40 << 30, dtype=torch.uint8, device=dev); del x; torch.cuda.empty_cache()
if mode.startswith("main"):
    import contextlib, io
    sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "20", "--repeats", "2"] + (["--no-extras"] if mode == "main_noextras" else [])
    if mode == "main_noc5":
        del bench.OTHER_CONFIGS[name]
    if mode == "main_only_c5":
        for k in list(bench.OTHER_CONFIGS):
            if k != name: del bench.OTHER_CONFIGS[k]
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        bench.main()
    if mode == "main_only_c5":
        import json
        d = json.loads(buf.getvalue().strip().splitlines()[-1])
        print(mode, "inside bench:", round(d["other_configs"][name]["scans_per_s"]))
        sys.exit(0)
    bench.OTHER_CONFIGS[name] = cfg
r = bench.run_other_config(name, cfg, capi, torch, dev, os.cpu_count() or 1, 0.02, -0.015, in_flight=6)
print(mode, round(r["scans_per_s"]), round(r["one_at_a_time"]["scans_per_s"]))
