#!/usr/bin/env bash
# forward-sweep ablation on workload $1 (default c2): libdgn_<variant>.so built with -DDGN_EXP_STAGE=n / -DDGN_EXP_NOSTORE
w=${1:-c2}; shift
for lib in libdgn_hip "$@"; do
  DGN_HIP_LIB=$PWD/dgn_amd/$lib.so python bench.py --workload $w --no-cpu-baseline --steps 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('$lib', {k: round(v['ms'], 4) for k, v in r['kernels'].items()})"
done
