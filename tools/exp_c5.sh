#!/usr/bin/env bash
# run the c5 workload at a given scale with each experimental library variant; print ms and kernel split
scale="${1:-0.25}"; shift || true
for lib in libdgn_hip "$@"; do
  DGN_HIP_LIB=$PWD/dgn_amd/$lib.so python bench.py --workload c5 --scale $scale --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('$lib', 'ms_per_step', round(d['ms_per_step'], 3), 'frac', round(r['frac'], 3), 'E', d['config']['edges_per_gpu'], 'n_hub', r['model']['n_hub'])"
done
