#!/usr/bin/env bash
# Build libdgn_hip.so (gfx950 only) in-tree: dgn_amd/libdgn_hip.so.  Incremental (csrc/Makefile: a translation unit is recompiled when
# it or one of its headers changed); `DGN_REBUILD=1` forces everything.
# Also fails the build if any kernel needs scratch (private) memory: accumulator arrays demoted to memory
# (dynamic indexing, switch lookup tables) silently halve the speed of these kernels.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
root="$(cd "$here/../.." && pwd)"
out="${DGN_OUT:-$root/dgn_amd/libdgn_hip.so}"
objdir="${DGN_OBJDIR:-$here}"
mkdir -p "$objdir"
[ "${DGN_REBUILD:-0}" = 1 ] && rm -f "$objdir"/*.o
# dependency files: the quoted includes of every unit, followed transitively through csrc/ and include/ (hipcc -MMD writes one
# bundle per offload target, which make cannot read)
deps_of() {   # $1 = file; prints the closure of its local includes
  local f="$1" inc p
  for inc in $(grep -h -o -E '^#include "[^"]+"' "$f" | sed -E 's/#include "(.*)"/\1/' | sort -u); do
    for p in "$here/$inc" "$root/include/$inc"; do
      if [ -f "$p" ] && ! grep -q -x -F "$p" "$seen"; then echo "$p" >> "$seen"; deps_of "$p"; fi
    done
  done
}
for f in "$here"/*.hip; do
  u="$(basename "$f" .hip)"; seen="$(mktemp)"; deps_of "$f"
  { printf '%s/%s.o:' "$objdir" "$u"; tr '\n' ' ' < "$seen"; echo; } > "$objdir/$u.d"; rm -f "$seen"
done
make -s -j "${DGN_JOBS:-$(nproc)}" -f "$here/Makefile" OUT="$out" OBJDIR="$objdir" ${HIPCC:+HIPCC="$HIPCC"}
grep -h -E "warning" "$objdir"/*.remarks | grep -v "Rpass" | head -20 || true
# (kernels of this library only: the rocPRIM sort / scan kernels dgn_graph_build.hip instantiates are the library's business)
scratch="$(grep -h -B8 -E "ScratchSize \[bytes/lane\]: [1-9]" "$objdir"/*.remarks | grep -E "Function Name" | grep -v -E "rocprim|hipcub" || true)"
if [ -n "$scratch" ]; then
  echo "ERROR: a kernel uses scratch memory:" >&2
  echo "$scratch" | head -10 >&2
  exit 1
fi
echo "built $out ($(cat "$objdir"/*.remarks | grep -c "Function Name") kernels, no scratch)"
