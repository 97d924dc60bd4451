timeout 900 python -m pytest tests/test_block_layer_gpu.py -x -q -m gpu 2>&1 | tail -3
for w in c2_b128 zinc_json_b128 c1_b128; do
  timeout 300 python bench.py --workload $w --steps 200 --warmup 30 --hipgraph --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$w captured ms',r['ms_per_step'])
"
done
python tools/blk_phases.py c2_b128 2>&1 | grep t_bwd
