#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
for v in 0 1; do TSDR_SPEC_DBG_X=$v; if [ $v = 1 ]; then export TSDR_SPEC_DBG=1; fi; python3 bench.py --spectra-only spectrum --steps 50 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('dbg=$v', d['spectra_only']['spectrum']['us_per_call'])"; done
