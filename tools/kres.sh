#!/bin/bash
# kernel resource usage (VGPRs, scratch, occupancy, LDS) of a HIP translation unit: tools/kres.sh file.hip [extra flags]
cd "$(dirname "$0")/../sub-cortical_segmentation_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c "$@" -o /tmp/kres.o 2>&1 | python3 -c "
import sys,re,subprocess
cur=None;d={}
for ln in sys.stdin:
    m=re.search(r'Function Name: (\S+)',ln)
    if m: cur=m.group(1); d={}
    for k,pat in [('V','VGPRs: (\d+)'),('A','AGPRs: (\d+)'),('scr','ScratchSize \[bytes/lane\]: (\d+)'),('occ','Occupancy \[waves/SIMD\]: (\d+)'),('lds','LDS Size \[bytes/block\]: (\d+)'),('S','SGPRs: (\d+)')]:
        m2=re.search(pat,ln)
        if m2 and cur: d[k]=m2.group(1)
    if 'LDS Size' in ln and cur:
        name=subprocess.run(['c++filt',cur],capture_output=True,text=True).stdout.strip()
        name=name.replace('scs::','').replace('void ','')
        print('%-120s'%name[:120], ' '.join('%s=%s'%kv for kv in d.items()))
"
