#!/usr/bin/env bash
# timing of the dense-layer workloads + the degree-class tests on the GPU box: tools/ab_dc.sh
run() { python bench.py --workload $1 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2', round(d['ms_per_step'],4))"; }
timeout 300 python -m pytest tests/test_dc_hip.py -x -q 2>&1 | tail -2
for w in c2c c1 c4 zinc_json; do run $w a; run $w b; done
