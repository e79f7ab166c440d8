#!/bin/bash
# Development aid: libtempest_hip.so with UBSan on the HOST side of every translation unit (trap mode: no runtime library, a
# hit kills the process at the spot; the device code is compiled as usual, at -O3: at -O1 the GPU suite takes over an hour --
# sanitizers do not exist for amdgcn) -> ab/ubsan_host.so.
# On a GPU box: TSDR_HIP_LIB=$PWD/ab/ubsan_host.so python -m pytest tests -m gpu; ... tools/fuzz_api_errors.py; tools/fuzz_*.py
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d); mkdir -p $R/ab
for f in $R/tempestsdr.jl_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -Xarch_device -O3 -g -ffp-contract=off -fPIC -std=c++17 -w -Xarch_host -fsanitize=undefined \
    -Xarch_host -fsanitize-trap=undefined -I$R/include -c "$f" -o $T/$(basename "$f" .hip).o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab/ubsan_host.so $T/*.o
rm -rf $T; ls -la $R/ab/ubsan_host.so
