#!/usr/bin/env bash
scale="${1:-0.25}"; shift
for lib in libdgn_hip "$@"; do
for cfg in "mean|identity" "mean max min sum std dir1-dx dir2-dx dir3-dx|identity amplification attenuation"; do
  a="${cfg%%|*}"; s="${cfg#*|}"
  DGN_HIP_LIB=$PWD/dgn_amd/$lib.so python bench.py --workload c5 --scale $scale --steps 3 --warmup 1 --aggregators "$a" --scalers "$s" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('$lib', '[$a]', 'ms', round(d['ms_per_step'], 3), 'GB/s', round(r['achieved']))"
done; done
