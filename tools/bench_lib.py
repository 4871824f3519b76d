"""Diagnostic: bench.py's headline measurement for an alternative build of the library under feature_extraction_amd/lib.
  python tools/bench_lib.py libfx_variant.so [bench args...]"""
import contextlib
import io
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_extraction_amd import capi

lib = sys.argv[1]
capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), lib)
import bench  # noqa: E402

sys.argv = ["bench.py", "--no-extras", "--no-cpu-baseline", "--check", "0", "--steps", "60"] + sys.argv[2:]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print(lib, round(d["value"]), round(d["ms_per_step"], 4), {k: round(v, 3) for k, v in d["kernel_ms"].items()})
