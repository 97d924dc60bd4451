#!/usr/bin/env bash
# On the GPU box: the degree-class tests, then step time and the three posttrans products of the dense-layer workloads, with the
# route on and off:  tools/ab_dc.sh
run() { python bench.py --workload $1 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print('$1', '$2', round(d['ms_per_step'],4), {n: round(v['ms'],4) for n,v in k.items() if n.startswith('dc_')})"; }
timeout 300 python -m pytest tests/test_dc_hip.py -x -q 2>&1 | tail -1
for w in c2c c1 c4 zinc_json; do run $w class-route; DGN_DC_POSTTRANS=0 run $w folded; done
