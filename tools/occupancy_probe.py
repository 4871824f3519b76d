"""Diagnostic: workgroups of k_rings_runs a CU holds as a function of its dynamic LDS (hipOccupancyMaxActiveBlocksPerMultiprocessor):
how LDS allocation is granulated on this device.  Run on the GPU box."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_extraction_amd import capi
lib = capi.load()
hip = C.CDLL("libamdhip64.so")
fn = C.cast(getattr(lib, "k_rings_runs"), C.c_void_p)
n = C.c_int()
prev = None
for lds in range(6 * 1024, 16 * 1024 + 1, 128):
    r = hip.hipOccupancyMaxActiveBlocksPerMultiprocessor(C.byref(n), fn, 64, C.c_size_t(lds))
    if n.value != prev:
        print("lds", lds, "-> blocks/CU", n.value, "rc", r)
        prev = n.value
