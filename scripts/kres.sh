#!/bin/bash
# Register / LDS / occupancy summary of every kernel of one .hip file (cross-compiles, no GPU needed).
#   scripts/kres.sh web-splat_amd/csrc/raster.hip [extra hipcc flags]
f=$1; shift
cd "$(dirname "$0")/.."
strict=""; case "$f" in *preprocess.hip|*ply_decode.hip) strict="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fopenmp -Iinclude -Iweb-splat_amd/csrc $strict "$@" -c "$f" -o /tmp/kres_$$.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import re,sys,subprocess
cur=None; rows=[]
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m:
        cur={'name':m.group(1)}; rows.append(cur); continue
    for k,pat in (('vgpr',r' VGPRs: (\d+)'),('agpr',r'AGPRs: (\d+)'),('sgpr',r' SGPRs: (\d+)'),('scratch',r'ScratchSize \[bytes/lane\]: (\d+)'),('occ',r'Occupancy \[waves/SIMD\]: (\d+)'),('lds',r'LDS Size \[bytes/block\]: (\d+)')):
        m=re.search(pat,line)
        if m and cur is not None: cur[k]=int(m.group(1))
names=subprocess.run(['c++filt']+[r['name'] for r in rows],capture_output=True,text=True).stdout.split('\n') if rows else []
for r,n in zip(rows,names):
    n=re.sub(r'ws::\(anonymous namespace\)::','',n); n=re.sub(r'\(.*','',n)
    print(f\"{n[:70]:70s} vgpr {r.get('vgpr',0):3d} agpr {r.get('agpr',0):3d} sgpr {r.get('sgpr',0):3d} scratch {r.get('scratch',0):4d} occ {r.get('occ',0)} lds {r.get('lds',0)}\")
"
rm -f /tmp/kres_$$.o
