#!/usr/bin/env bash
# step time of one workload under several environment settings, interleaved:  tools/ab_step.sh "<bench args>" "VAR=v ..." "VAR=v ..." ...
args="$1"; shift
for rep in 1 2 3; do for setting in "$@"; do
  env $setting python bench.py --no-extras --no-cpu-baseline $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$setting', round(d['ms_per_step'],4))"
done; done
