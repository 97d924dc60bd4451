#!/usr/bin/env bash
run() { python bench.py --workload $1 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4))"; }
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for w in c2c c1 c4 c2; do run $w; done
