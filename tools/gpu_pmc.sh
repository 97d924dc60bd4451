#!/usr/bin/env bash
# Usage: tools/gpu_pmc.sh <tag> "<counters>" <bench args...>   (one PMC pass, kernel-trace only; at most 4 counters of one
# block per pass -- an over-subscribed request aborts rocprofv3 and then hangs in its signal handler, hence the timeout)
set -uo pipefail
tag="$1"; ctrs="$2"; shift 2
export TMPDIR=/tmp
out="gpurun_out/pmc_$tag"
mkdir -p "$out"
timeout -k 5 240 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out" -o "$tag" -- python bench.py "$@" > "$out/log.txt" 2>&1
python tools/pmc_summary.py "$out/${tag}_counter_collection.csv" | grep -v "^ew_"
