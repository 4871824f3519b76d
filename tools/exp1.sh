run() { echo "== $*"; python tools/bench_lib.py "$@" 2>/dev/null | tail -1; }
python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_front.py tests/test_gpu_configs.py -x -q 2>&1 | tail -2
run libfx_hip.so
run libfx_hip.so
run libfx_hip.so --contexts 1
FX_FRONT=0 run libfx_hip_test.so
FX_FRONT=0 run libfx_hip_test.so --contexts 1
