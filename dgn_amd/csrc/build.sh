#!/usr/bin/env bash
# Build libdgn_hip.so (gfx950 only) in-tree: dgn_amd/libdgn_hip.so
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
root="$(cd "$here/../.." && pwd)"
out="$root/dgn_amd/libdgn_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$root/include" -I"$here" -Wall -Wno-unused-function -munsafe-fp-atomics -ffp-contract=off)
objs=()
for f in dgn_abi dgn_edge_weights dgn_agg; do
  "$HIPCC" "${FLAGS[@]}" -c "$here/$f.hip" -o "$here/$f.o" &
  objs+=("$here/$f.o")
done
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$out"
echo "built $out"
