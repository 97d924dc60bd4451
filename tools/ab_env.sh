#!/usr/bin/env bash
# A/B of an environment switch on the sweep legs of one workload:  tools/ab_env.sh VAR [bench args...]   (A = unset, B = VAR=1)
var="$1"; shift
for v in A B A B; do
  if [ $v = A ]; then unset $var; else export $var=1; fi
  python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$v', 'step', round(d['ms_per_step'],4), 'fwd', round(k['agg_fwd_rows']['ms'],4), round(k['agg_fwd_rows']['frac'],3), 'bwd', round(k['agg_bwd_rows']['ms'],4), round(k['agg_bwd_rows']['frac'],3))"
done
