#!/bin/bash
# usage: tools/kernel_resources.sh file.hip  -> kernel, VGPRs, scratch bytes/lane, occupancy
f=$1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast -fPIC -std=c++17 -c $f -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
name=None; d={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: name=m.group(1); d[name]={}
    for k in ['VGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]','VGPRs Spill']:
        m=re.search(r'remark:\s+'+k+r': (\d+)',l)
        if m and name: d[name][k]=m.group(1)
for n,v in d.items():
    dn=subprocess.run(['c++filt',n],capture_output=True,text=True).stdout.strip()[:70]
    print(f\"{dn:70s} vgpr {v.get('VGPRs','?'):>4s} scratch {v.get('ScratchSize \\\\[bytes/lane\\\\]','?'):>4s} occ {v.get('Occupancy \\\\[waves/SIMD\\\\]','?')} spill {v.get('VGPRs Spill','?')}\")
"
