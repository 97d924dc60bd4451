#!/usr/bin/env bash
# Usage (on the GPU box, from the repo root):  tools/gpu_profile.sh <tag> <bench args...>
# Pass 1: timeout -k 5 600 rocprofv3 --kernel-trace --stats.  Pass 2/3: PMC FETCH_SIZE / WRITE_SIZE (separate passes, kernel-trace only).
set -uo pipefail
tag="$1"; shift
export TMPDIR=/tmp
out="gpurun_out/prof_$tag"
mkdir -p "$out"
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o "$tag" -- python bench.py "$@" > "$out/trace.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc_$c" -o "$tag" -- python bench.py "$@" > "$out/pmc_$c.log" 2>&1
done
ls -R "$out" | head -30
