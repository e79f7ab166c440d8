#!/bin/bash
# Development aid: build ab/<name>.so from the working tree with one sed edit applied to one csrc file (e.g. a header) and
# one or more translation units recompiled.   tools/mk_variant2.sh <name> <edited file> '<sed expression>' <unit.hip> [unit.hip ...]
set -e
NAME=$1; FILE=$2; EXPR=$3; shift 3
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
mkdir -p "$TMP/tempestsdr.jl_amd" "$TMP/obj" "$ROOT/ab"
cp -r "$ROOT/tempestsdr.jl_amd/csrc" "$TMP/tempestsdr.jl_amd/"; cp -r "$ROOT/include" "$TMP/"
sed -i "$EXPR" "$TMP/tempestsdr.jl_amd/csrc/$FILE"
if cmp -s "$TMP/tempestsdr.jl_amd/csrc/$FILE" "$ROOT/tempestsdr.jl_amd/csrc/$FILE"; then echo "sed changed nothing"; exit 1; fi
cp "$ROOT"/tempestsdr.jl_amd/build/*.o "$TMP/obj/"
for u in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -w $EXTRA_FLAGS -c "$TMP/tempestsdr.jl_amd/csrc/$u" -o "$TMP/obj/$(basename "$u" .hip).o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/ab/$NAME.so" "$TMP"/obj/*.o -L/opt/rocm/lib -lrccl
rm -rf "$TMP"; ls -la "$ROOT/ab/$NAME.so"
