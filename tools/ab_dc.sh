#!/usr/bin/env bash
run() { python bench.py --workload $1 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print('$1', '$2', round(d['ms_per_step'],4), {n: round(v['ms'],4) for n,v in k.items() if n.startswith('dc_')})"; }
timeout 300 python -m pytest tests/test_dc_hip.py -x -q 2>&1 | tail -1
for w in c2c c4 c1; do run $w prio; DGN_DC_WPRIO=1 run $w wprio; DGN_DC_ABL=8 run $w noprio; done
