#!/bin/bash
# CPU only: the oracle (test infrastructure) under UBSan and ASan through its own CPU tests.  The GPU library itself cannot be
# sanitized on this pool (no GPU ASan / XNACK); its C ABI is exercised by tools/fuzz_api_errors.py instead.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/oracle
cp libtempest_oracle.so /tmp/liborc_backup.so
restore() { cp /tmp/liborc_backup.so $R/oracle/libtempest_oracle.so; touch $R/oracle/libtempest_oracle.so $R/oracle/libtempest_oracle_alt.so; }
trap restore EXIT
T="tests/test_oracle_pins.py tests/test_golden.py tests/test_julia_golden.py tests/test_replay.py"
gcc -O1 -g -fPIC -std=gnu11 -ffp-contract=off -fno-fast-math -fsanitize=undefined -fno-sanitize-recover=undefined -shared -o libtempest_oracle.so tempest_oracle.c -lm -lubsan
(cd $R && python -m pytest $T -x -q -m "not gpu" | tail -1)
gcc -O1 -g -fPIC -std=gnu11 -ffp-contract=off -fno-fast-math -fsanitize=address -shared -o libtempest_oracle.so tempest_oracle.c -lm
(cd $R && ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so) python -m pytest $T -x -q -m "not gpu" | tail -1)
