#!/usr/bin/env bash
# sweep legs of one workload under several values of an environment variable:  tools/ab_val.sh VAR "v1 v2 ..." [bench args...]  ("-" = unset)
var="$1"; vals="$2"; shift 2
for rep in 1 2; do for v in $vals; do
  if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
  python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$var=$v', 'step', round(d['ms_per_step'],4), 'fwd', round(k['agg_fwd_rows']['ms'],4), round(k['agg_fwd_rows']['frac'],3), 'bwd', round(k['agg_bwd_rows']['ms'],4), round(k['agg_bwd_rows']['frac'],3))"
done; done
