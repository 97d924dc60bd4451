#!/usr/bin/env bash
# timing of the dense-layer workloads + the degree-class tests on the GPU box: tools/ab_dc.sh
run() { python bench.py --workload $1 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print('$1', '$2', round(d['ms_per_step'],4), {n: round(v['ms'],4) for n,v in k.items() if n.startswith('dc_')})"; }
DGN_DC_DEEP=1 timeout 300 python -m pytest tests/test_dc_hip.py -x -q 2>&1 | tail -2
for w in c2c c1 c4 zinc_json; do DGN_DC_DEEP=1 run $w deep; run $w base; done
