#!/usr/bin/env bash
scale="${1:-0.25}"
run() { python bench.py --workload c5 --scale $scale --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('  ms', round(d['ms_per_step'], 3), 'GB/s', round(r['achieved']), 'bytes', r['kernels']['agg_fwd(all launches)']['bytes'])"; }
for mode in rows; do
  if [ $mode = rows ]; then export DGN_FWD_ROWS=1; else unset DGN_FWD_ROWS; fi
  echo "== $mode: mean / identity"; run --aggregators "mean" --scalers identity
  echo "== $mode: mean dir1-dx / identity"; run --aggregators "mean dir1-dx" --scalers identity
  echo "== $mode: 8 aggs / identity"; run --scalers identity
  echo "== $mode: 8 aggs / 3 scalers"; run
done
