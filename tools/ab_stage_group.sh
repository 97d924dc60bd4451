for v in A B A B; do
  if [ $v = A ]; then unset DGN_NO_STAGE_GROUP; else export DGN_NO_STAGE_GROUP=1; fi
  python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$v', 'step', round(d['ms_per_step'],4), 'fwd', round(k['agg_fwd_rows']['ms'],4), k['agg_fwd_rows']['frac'], 'bwd', round(k['agg_bwd_rows']['ms'],4))"
done
