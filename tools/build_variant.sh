#!/bin/bash
# Development aid: build libtempest_hip.so of another git revision into ab/<name>.so for same-box A/B timing
# (bench.py loads it when TSDR_HIP_LIB points at it).   tools/build_variant.sh <git-ref> <name>
set -e
REF=$1; NAME=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
git -C "$ROOT" archive "$REF" tempestsdr.jl_amd/csrc include | tar -x -C "$TMP"
mkdir -p "$ROOT/ab" "$TMP/obj"
for f in "$TMP"/tempestsdr.jl_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -w -c "$f" -o "$TMP/obj/$(basename "$f" .hip).o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/ab/$NAME.so" "$TMP"/obj/*.o -L/opt/rocm/lib -lrccl
rm -rf "$TMP"
ls -la "$ROOT/ab/$NAME.so"
