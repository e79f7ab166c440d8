#!/bin/bash
# Development aid: build the WORKING TREE's library with extra compiler flags into ab/<name>.so (every file recompiled: for
# macros that headers shared by several files read).   tools/build_flags.sh <name> -DFLAG=...
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d); mkdir -p "$ROOT/ab" "$TMP/obj"
for f in "$ROOT"/tempestsdr.jl_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -w -I"$ROOT/include" "$@" -c "$f" -o "$TMP/obj/$(basename "$f" .hip).o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/ab/$NAME.so" "$TMP"/obj/*.o -L/opt/rocm/lib -lrccl
rm -rf "$TMP"; ls -la "$ROOT/ab/$NAME.so"
