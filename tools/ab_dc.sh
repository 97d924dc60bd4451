#!/usr/bin/env bash
# A/B of the degree-class posttrans switches on the GPU box: tools/ab_dc.sh
run() { python bench.py --workload $1 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2', round(d['ms_per_step'],4))"; }
timeout 300 python -m pytest tests/test_dc_hip.py tests/test_abi.py -x -q 2>&1 | tail -3
for i in 1 2; do run c2 dc_towers; DGN_DC_TOWERS=0 run c2 folded; done
