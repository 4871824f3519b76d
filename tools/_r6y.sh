#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for r in 1 2 3; do for b in 1 8; do python3 tools/latency.py $b 300 launch 0 2>/dev/null | tail -2 | head -1 | cut -c1-260; done; done
timeout 1800 python3 -m pytest tests -m gpu -x -q > gpurun_out/r6y9_tests.log 2>&1; grep -n "passed\|failed" gpurun_out/r6y9_tests.log
python3 bench.py --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j.get('host_to_host_scans_per_s'), j.get('h2d_inclusive_scans_per_s'))"
