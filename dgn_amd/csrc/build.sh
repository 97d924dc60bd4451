#!/usr/bin/env bash
# Build libdgn_hip.so (gfx950 only) in-tree: dgn_amd/libdgn_hip.so
# Also fails the build if any kernel needs scratch (private) memory: accumulator arrays demoted to memory
# (dynamic indexing, switch lookup tables) silently halve the speed of these kernels.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
root="$(cd "$here/../.." && pwd)"
out="${DGN_OUT:-$root/dgn_amd/libdgn_hip.so}"
objdir="${DGN_OBJDIR:-$here}"
mkdir -p "$objdir"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$root/include" -I"$here" -Wall -Wno-unused-function
       -munsafe-fp-atomics -ffp-contract=off -Rpass-analysis=kernel-resource-usage ${DGN_EXTRA_FLAGS:-})
objs=()
pids=()
for f in dgn_abi dgn_towers dgn_gemm dgn_fused dgn_graph_build dgn_edge_weights dgn_combine dgn_bn_tail dgn_linear dgn_linear_bn dgn_linear_act dgn_linear_add dgn_linear_mix dgn_linear_combine dgn_linear_expand dgn_linear_wgrad dgn_agg dgn_agg_v1 dgn_agg_v2 dgn_agg_v4; do
  ( "$HIPCC" "${FLAGS[@]}" -c "$here/$f.hip" -o "$objdir/$f.o" 2> "$objdir/$f.remarks" ) &
  pids+=($!)
  objs+=("$objdir/$f.o")
done
fail=0
for pid in "${pids[@]}"; do wait "$pid" || fail=1; done
if [ "$fail" != 0 ]; then grep -h -E "error|Error" "$objdir"/*.remarks | head -40; exit 1; fi
grep -h -E "warning" "$objdir"/*.remarks | grep -v "Rpass" | head -20 || true
# (kernels of this library only: the rocPRIM sort / scan kernels dgn_graph_build.hip instantiates are the library's business)
scratch="$(grep -h -B8 -E "ScratchSize \[bytes/lane\]: [1-9]" "$objdir"/*.remarks | grep -E "Function Name" | grep -v -E "rocprim|hipcub" || true)"
if [ -n "$scratch" ]; then
  echo "ERROR: a kernel uses scratch memory:" >&2
  echo "$scratch" | head -10 >&2
  exit 1
fi
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$out"
echo "built $out ($(cat "$objdir"/*.remarks | grep -c "Function Name") kernels, no scratch)"
