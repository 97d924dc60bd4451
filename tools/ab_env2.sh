#!/usr/bin/env bash
# backward sweep of workloads under env settings: tools/ab_env2.sh "w1 w2" "SET1" "SET2" ...
ws="$1"; shift
for rep in 1 2; do for setting in "$@"; do for w in $ws; do
  env $setting python bench.py --workload $w --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$setting', '$w', 'step', round(d['ms_per_step'],4), 'fwd', round(k['agg_fwd_rows']['ms'],4), 'bwd', round(k['agg_bwd_rows']['ms'],4), round(k['agg_bwd_rows']['frac'],3))"
done; done; done
