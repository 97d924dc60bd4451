#!/usr/bin/env bash
# A library variant for A/B runs: tools/variant.sh NAME "<extra hipcc flags>" unit [unit ...]
# recompiles the named translation units with the extra flags into dgn_amd/csrc/_variants/NAME/ and links them with the main build's
# other objects into dgn_amd/libdgn_hip_NAME.so (select it with DGN_HIP_LIB=$PWD/dgn_amd/libdgn_hip_NAME.so).
set -euo pipefail
name="$1"; flags="$2"; shift 2
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; here="$root/dgn_amd/csrc"; vdir="$here/_variants/$name"
mkdir -p "$vdir"
objs=()
for o in "$here"/*.o; do
  u="$(basename "$o" .o)"; skip=0
  for v in "$@"; do [ "$u" = "$v" ] && skip=1; done
  [ $skip = 0 ] && objs+=("$o")
done
for u in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$root/include" -I"$here" -Wno-unused-function -munsafe-fp-atomics \
    -ffp-contract=off $flags -c "$here/$u.hip" -o "$vdir/$u.o" 2> "$vdir/$u.log" &
done
wait
for u in "$@"; do objs+=("$vdir/$u.o"); done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$root/dgn_amd/libdgn_hip_$name.so"
echo "built dgn_amd/libdgn_hip_$name.so"
