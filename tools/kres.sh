#!/bin/bash
# Development aid: VGPR / scratch / LDS per kernel of one .hip file.   tools/kres.sh <file.hip> [filter]
F=$1; PAT=${2:-.}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -w -I$ROOT/include -c "$F" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
cur=None;rows=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur={'n':m.group(1)}; rows.append(cur); continue
    for k,pat in (('v',r' VGPRs: (\d+)'),('a',r'AGPRs: (\d+)'),('s',r'ScratchSize \[bytes/lane\]: (\d+)'),('o',r'Occupancy \[waves/SIMD\]: (\d+)'),('l',r'LDS Size \[bytes/block\]: (\d+)')):
        m=re.search(pat,l)
        if m and cur is not None: cur[k]=m.group(1)
names=subprocess.run(['c++filt']+[r['n'] for r in rows],capture_output=True,text=True).stdout.split('\n')
for r,n in zip(rows,names):
    n=re.sub(r'\(.*','',n).replace('void ','').replace('tsdr::','')
    if re.search(r'''$PAT''',n): print(f\"{n:48s} vgpr {r.get('v'):>4} scratch {r.get('s'):>4} occ {r.get('o'):>2} lds {r.get('l')}\")
"
