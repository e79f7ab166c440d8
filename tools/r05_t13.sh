#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
cd $R
timeout 900 python3 -m pytest tests/test_fft_path_gpu.py -x -q -m gpu -k "spectrum" 2>&1 | tail -6
for v in 1 0 1 0; do TSDR_SPECTRUM_ONE=$v python3 bench.py --spectra-only spectrum --steps 50 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('spectrum_one=$v', d['spectra']['spectrum']['us_per_call'] if 'spectra' in d else d)"; done
