#!/usr/bin/env bash
# per-kernel rocprofv3 --stats table of one workload for the main library and a variant:  tools/stats_ab.sh <variant> <bench args...>
set -uo pipefail
var="$1"; shift
export TMPDIR=/tmp
for lib in libdgn_hip "libdgn_hip_$var"; do
  out="gpurun_out/stats_$lib"; rm -rf "$out"; mkdir -p "$out"
  DGN_HIP_LIB=$PWD/dgn_amd/$lib.so timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o x -- python bench.py "$@" > "$out/log.txt" 2>&1
  echo "== $lib"; python tools/prof_summary.py "$(find $out -name '*kernel_stats.csv' | head -1)" 24
done
