for w in "$@"; do
  timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); k=r['roofline']['kernels']
        print('$w','ms',r['ms_per_step'],' '.join('%s %.4f %.3f'%(n,v['ms'],v['frac']) for n,v in k.items()))
"
done
