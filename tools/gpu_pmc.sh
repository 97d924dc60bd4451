#!/usr/bin/env bash
# Usage: tools/gpu_pmc.sh <tag> "<counters>" <bench args...>   (one PMC pass, kernel-trace only)
set -uo pipefail
tag="$1"; ctrs="$2"; shift 2
export TMPDIR=/tmp
out="gpurun_out/pmc_$tag"
mkdir -p "$out"
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out" -o "$tag" -- python bench.py "$@" > "$out/log.txt" 2>&1
python tools/pmc_summary.py "$out/${tag}_counter_collection.csv" | grep -v "^ew_"
