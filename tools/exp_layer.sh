#!/usr/bin/env bash
# kernel timings of the layer workloads for each library variant
for lib in libdgn_hip "$@"; do
for w in c2 c4 c1 c3; do
  DGN_HIP_LIB=$PWD/dgn_amd/$lib.so python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('$lib', '$w', 'step_ms', round(d['ms_per_step'], 3), {k: (round(v['ms'], 4), round(v.get('GBps', 0))) for k, v in r['kernels'].items()})"
done; done
