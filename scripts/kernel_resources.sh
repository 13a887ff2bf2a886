#!/bin/bash
# VGPR / SGPR / scratch / occupancy per kernel of one .hip source (compiler view; usage: kernel_resources.sh m3d_fftconv.hip [extra flags])
cd "$(dirname "$0")/../merfish3d-analysis_amd/csrc"
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC "$@" -Rpass-analysis=kernel-resource-usage -c -o /tmp/kres.o $src 2>&1 | python3 -c "
import sys,re,subprocess
name=None; rows={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: name=m.group(1); rows[name]={}
    for key in ['VGPRs','AGPRs','ScratchSize','Occupancy','SGPRs','LDS Size','VGPRs Spill','SGPRs Spill']:
        m=re.search(r' '+key+r'( \[[^\]]*\])?: (\d+)',l)
        if m and name: rows[name].setdefault(key,m.group(2))
for n,r in rows.items():
    d=subprocess.run(['c++filt',n],capture_output=True,text=True).stdout.strip()
    d=d.replace('(anonymous namespace)::','').replace('void ','')
    d=re.sub(r'\(.*','',d)
    print(f'{d:45s}',' '.join(f'{k}={v}' for k,v in r.items()))
"
